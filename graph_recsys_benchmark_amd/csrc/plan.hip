// Graph plan: the one-time preprocessing of the P x S list of COO edge_index tensors the reference
// builds in graph_recsys_benchmark/utils/general_utils.py:280-395 (int64 [2,E], row 0 = source,
// row 1 = target, multi-edges kept) into what the gfx950 kernels consume:
//   - destination-sorted CSR with int32 ids, STABLE (edge order inside a row = COO order), optionally
//     with existing self loops dropped (GAT: remove_self_loops + add_self_loops, GCN:
//     add_remaining_self_loops; the one-loop-per-node is added inside the kernels, never materialised)
//   - degree bins: short rows (row per lane subgroup), long rows (row per wave), hub rows cut into
//     <= 512-edge chunks (destination-degree skew)
//   - GCN deg^-1/2 (computed once; the reference recomputes it every call)
// The reference rebuilds the self-loop edge list and the GCN norm on every forward; here it is static.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace pea {
namespace {

__global__ void coo_to_keys(int64_t E, int64_t N, const int64_t *__restrict__ coo, int drop_loops,
                            int *__restrict__ keys, int *__restrict__ vals, int *err) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t j = coo[e], i = coo[E + e];
    if (j < 0 || j >= N || i < 0 || i >= N) {
        atomicOr(err, 1);
        keys[e] = (int)N;
        vals[e] = 0;
        return;
    }
    keys[e] = (drop_loops && i == j) ? (int)N : (int)i;  // dropped edges sort behind every real row
    vals[e] = (int)j;
}

// (dst, source-slice) keys: inside a destination row the edges are grouped by which slice of the source-id range
// they gather from, so a workgroup can be handed edges whose source rows share one XCD's L2
__global__ void keys_with_slice(int64_t E, int64_t N, const int *__restrict__ keys, const int *__restrict__ vals, int smin,
                                int64_t span, int S, unsigned long long *__restrict__ keys64) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int k = keys[e];
    int sl = 0;
    if (k < (int)N) sl = (int)(((int64_t)(vals[e] - smin) * S) / span);
    keys64[e] = (unsigned long long)k * (unsigned)S + (unsigned)sl;
}

__global__ void src_range(int64_t E, int64_t N, const int *__restrict__ keys, const int *__restrict__ vals, int *mn, int *mx) {
    int lo = INT32_MAX, hi = -1;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x)
        if (keys[e] < (int)N) {
            lo = min(lo, vals[e]);
            hi = max(hi, vals[e]);
        }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0 && hi >= 0) {  // one atomic pair per wave
        atomicMin(mn, lo);
        atomicMax(mx, hi);
    }
}

__global__ void rowptr_from_sorted64(int64_t E, int64_t N, int S, const unsigned long long *__restrict__ keys,
                                     int *__restrict__ rowptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    const unsigned long long want = (unsigned long long)i * (unsigned)S;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    rowptr[i] = (int)lo;
}

// bounds[h*(S+1) + s] = first edge of hub row rows[h] whose slice is >= s
__global__ void slice_bounds(int n_hub, int S, int64_t E, const int *__restrict__ rows, const unsigned long long *__restrict__ keys,
                             int *__restrict__ bounds) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_hub * (S + 1)) return;
    const int h = t / (S + 1), sl = t % (S + 1);
    const unsigned long long want = (unsigned long long)rows[h] * (unsigned)S + (unsigned)sl;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    bounds[t] = (int)lo;
}

// rowptr[i] = first position whose key >= i  (i = 0..N)
__global__ void rowptr_from_sorted(int64_t E, int64_t N, const int *__restrict__ keys, int *__restrict__ rowptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < (int)i) lo = mid + 1;
        else hi = mid;
    }
    rowptr[i] = (int)lo;
}

__global__ void iota_kernel(int64_t n, int *p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int)i;
}

__global__ void gather_src(int64_t n, const int *__restrict__ eid, const int *__restrict__ src, int *__restrict__ col) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) col[i] = src[eid[i]];
}

__global__ void count_sources(int64_t E, const int *__restrict__ col, int *__restrict__ cnt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(&cnt[col[e]], 1);
}

__global__ void dinv_from_counts(int64_t N, const int *__restrict__ cnt, const int *__restrict__ rowptr,
                                 int from_col, int self_loops, float *__restrict__ dinv) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int d = (from_col ? rowptr[v + 1] - rowptr[v] : cnt[v]) + (self_loops ? 1 : 0);
    const float r = powf((float)d, -0.5f);
    dinv[v] = isinf(r) ? 0.f : r;
}

template <typename T>
int upload(const std::vector<T> &h, T **d) {
    *d = nullptr;
    if (h.empty()) return PEA_OK;
    PEA_HIP(hipMalloc((void **)d, h.size() * sizeof(T)));
    PEA_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return PEA_OK;
}

int build_relation(pea_plan *plan, Relation &R, const int64_t *coo_dev, int64_t E, hipStream_t stream) {
    const int64_t N = plan->N;
    const bool drop = plan->flags & PEA_PLAN_SELF_LOOPS;
    R.e_in = E;
    PEA_REQUIRE(E >= 0 && E < (int64_t)INT32_MAX - 64, PEA_ERR_ARG, "relation with %lld edges exceeds the int32 CSR", (long long)E);
    PEA_HIP(hipMalloc((void **)&R.rowptr, (size_t)(N + 1) * sizeof(int)));
    int *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr, *err = nullptr;
    unsigned long long *k64_in = nullptr, *k64_out = nullptr;
    int *hub_dev = nullptr, *bounds_dev = nullptr, *ids_in = nullptr;
    int S = 1;  // source slices (1 = edge order inside a row is plain COO order)
    void *tmp = nullptr;
    int rc = PEA_OK;
    const size_t eb = (size_t)std::max<int64_t>(E, 1) * sizeof(int);
    auto cleanup = [&]() {
        (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(vals_in); (void)hipFree(err); (void)hipFree(tmp);
        (void)hipFree(k64_in); (void)hipFree(k64_out); (void)hipFree(hub_dev); (void)hipFree(bounds_dev); (void)hipFree(ids_in);
        keys_in = keys_out = vals_in = err = hub_dev = bounds_dev = ids_in = nullptr;
        k64_in = k64_out = nullptr;
        tmp = nullptr;
    };
#define PEA_HIP_C(call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));          \
            cleanup();                                                                                    \
            return PEA_ERR_HIP;                                                                           \
        }                                                                                                 \
    } while (0)
    PEA_HIP_C(hipMalloc((void **)&keys_in, eb));
    PEA_HIP_C(hipMalloc((void **)&keys_out, eb));
    PEA_HIP_C(hipMalloc((void **)&vals_in, eb));
    PEA_HIP_C(hipMalloc((void **)&R.col, eb));
    PEA_HIP_C(hipMalloc((void **)&err, sizeof(int)));
    PEA_HIP_C(hipMemsetAsync(err, 0, sizeof(int), stream));
    if (E > 0) {
        hipLaunchKernelGGL(coo_to_keys, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, E, N, coo_dev,
                           drop ? 1 : 0, keys_in, vals_in, err);
        PEA_HIP_C(hipGetLastError());
        int bits = 1;
        while ((1ll << bits) <= N) ++bits;  // keys are in [0, N]
        // Large relations whose source rows overflow an XCD's 4 MiB L2: group each row's edges by source slice
        // (8 XCDs x phases), SURVEY.md section 7 "gather granularity" / guide: run workgroups that share rows on one XCD
        // (PEA_SLICE_MIN_EDGES / PEA_SLICE_BYTES: test knobs so small graphs can exercise the sliced layout)
        const char *env_e = getenv("PEA_SLICE_MIN_EDGES"), *env_b = getenv("PEA_SLICE_BYTES");
        const int64_t min_edges = env_e ? atoll(env_e) : kSliceMinEdges;
        const double slice_bytes = env_b ? atof(env_b) : kSliceBytes;
        {
            // source-id range of the kept edges: the footprint of the gather table (always recorded), and for large
            // relations the span the source slices divide
            int h_mm[2] = {INT32_MAX, -1}, *mm = nullptr;
            PEA_HIP_C(hipMalloc((void **)&mm, 2 * sizeof(int)));
            PEA_HIP_C(hipMemcpyAsync(mm, h_mm, sizeof(h_mm), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(src_range, dim3((unsigned)std::min<int64_t>((E + 255) / 256, 2048)), dim3(256), 0, stream, E, N, keys_in, vals_in, mm, mm + 1);
            PEA_HIP_C(hipMemcpyAsync(h_mm, mm, sizeof(h_mm), hipMemcpyDeviceToHost, stream));
            PEA_HIP_C(hipStreamSynchronize(stream));
            (void)hipFree(mm);
            if (h_mm[1] >= h_mm[0]) {
                const int64_t span = (int64_t)h_mm[1] - h_mm[0] + 1;
                R.src_span = span;
                const double footprint = (double)span * plan->gather_row_bytes;
                if (plan->gather_row_bytes > 0 && E >= min_edges && footprint > 1.5 * slice_bytes) {
                    const int phases = (int)std::min<double>(8.0, std::ceil(footprint / (8.0 * slice_bytes)));
                    S = 8 * phases;
                    R.slice_min = h_mm[0];
                    R.slice_span = span;
                }
            }
        }
        // with PEA_PLAN_EDGE_IDS the sort carries the ORIGINAL edge index (per-edge weights stay in COO order); the
        // source ids are gathered through it afterwards
        int *sort_vals = vals_in, *sort_out = R.col;
        if (plan->flags & PEA_PLAN_EDGE_IDS) {
            PEA_HIP_C(hipMalloc((void **)&ids_in, eb));
            PEA_HIP_C(hipMalloc((void **)&R.eid, eb));
            hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, E, ids_in);
            sort_vals = ids_in;
            sort_out = R.eid;
        }
        size_t tmp_bytes = 0;
        if (S == 1) {
            PEA_HIP_C(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, sort_vals, sort_out, (size_t)E, 0u,
                                                (unsigned)bits, stream));
            PEA_HIP_C(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
            PEA_HIP_C(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, sort_vals, sort_out, (size_t)E, 0u,
                                                (unsigned)bits, stream));
        } else {
            PEA_HIP_C(hipMalloc((void **)&k64_in, (size_t)E * sizeof(unsigned long long)));
            PEA_HIP_C(hipMalloc((void **)&k64_out, (size_t)E * sizeof(unsigned long long)));
            hipLaunchKernelGGL(keys_with_slice, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, E, N, keys_in, vals_in,
                               R.slice_min, R.slice_span, S, k64_in);
            PEA_HIP_C(hipGetLastError());
            int sbits = 1;
            while ((1 << sbits) < S) ++sbits;
            PEA_HIP_C(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k64_in, k64_out, sort_vals, sort_out, (size_t)E, 0u,
                                                (unsigned)(bits + sbits + 1), stream));
            PEA_HIP_C(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
            PEA_HIP_C(rocprim::radix_sort_pairs(tmp, tmp_bytes, k64_in, k64_out, sort_vals, sort_out, (size_t)E, 0u,
                                                (unsigned)(bits + sbits + 1), stream));
        }
        if (plan->flags & PEA_PLAN_EDGE_IDS) {
            hipLaunchKernelGGL(gather_src, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, E, R.eid, vals_in, R.col);
            PEA_HIP_C(hipGetLastError());
        }
    }
    if (S == 1)
        hipLaunchKernelGGL(rowptr_from_sorted, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, stream, E, N, keys_out,
                           R.rowptr);
    else
        hipLaunchKernelGGL(rowptr_from_sorted64, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, stream, E, N, S, k64_out,
                           R.rowptr);
    PEA_HIP_C(hipGetLastError());
    int herr = 0;
    PEA_HIP_C(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, stream));
    std::vector<int> rp((size_t)N + 1);
    PEA_HIP_C(hipMemcpyAsync(rp.data(), R.rowptr, (size_t)(N + 1) * sizeof(int), hipMemcpyDeviceToHost, stream));
    PEA_HIP_C(hipStreamSynchronize(stream));
    if (herr != 0) cleanup();
    PEA_REQUIRE(herr == 0, PEA_ERR_RANGE, "edge_index holds a node id outside [0, %lld)", (long long)N);
    R.e_kept = rp[(size_t)N];
    R.slices = S;

    // ---- degree bins over the rows this rank owns (host; rowptr only) ----
    std::vector<int> zero_rows, short_rows, hub_rows, hub_first, hub_count, sliced_rows;
    std::vector<LongItem> items;
    const int tile = plan->shard_tile, world = plan->shard_world, rank = plan->shard_rank;
    int max_deg = 0, slots = 0;
    const int short_deg = getenv("PEA_SHORT_DEG") ? atoi(getenv("PEA_SHORT_DEG")) : kShortDeg;  // tuning knobs
    const int chunk = getenv("PEA_CHUNK") ? atoi(getenv("PEA_CHUNK")) : kChunk;
    for (int64_t i = 0; i < N; ++i) {
        const int deg = rp[(size_t)i + 1] - rp[(size_t)i];
        max_deg = std::max(max_deg, deg);
        if (world > 1 && (int)((i / tile) % world) != rank) continue;
        R.rows_owned++;
        R.edges_owned += deg;
        if (deg == 0) {
            zero_rows.push_back((int)i);
        } else if (deg <= short_deg) {
            short_rows.push_back((int)i);
            R.edges_short += deg;
        } else if (deg <= chunk) {
            items.push_back({(int)i, rp[(size_t)i], rp[(size_t)i + 1], -1});
        } else if (S > 1 && deg >= S * kSliceMinSegment) {
            sliced_rows.push_back((int)i);  // chunked per source slice below
        } else {
            const int nch = (deg + chunk - 1) / chunk;
            const int len = (deg + nch - 1) / nch;
            hub_rows.push_back((int)i);
            hub_first.push_back(slots);
            hub_count.push_back(nch);
            for (int c = 0; c < nch; ++c) {
                const int b = rp[(size_t)i] + c * len;
                const int e = std::min(b + len, rp[(size_t)i + 1]);
                items.push_back({(int)i, b, e, slots++});
            }
        }
    }
    // longest first: the tail of the launch is made of the cheapest items
    std::stable_sort(items.begin(), items.end(),
                     [](const LongItem &a, const LongItem &b) { return (a.end - a.beg) > (b.end - b.beg); });
    if (!sliced_rows.empty()) {
        // per-slice segments of the big hub rows -> chunks, laid out so that workgroup b (4 items) of the launch
        // works on slice  phase*8 + (b % 8): workgroups are dealt round-robin over the 8 XCDs, so all gathers of
        // one slice meet in one XCD's L2 (placement affects speed only, never results)
        const int nh = (int)sliced_rows.size();
        std::vector<int> bounds((size_t)nh * (S + 1));
#define PEA_HIP_C2(call)                                                                                  \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));          \
            cleanup();                                                                                    \
            return PEA_ERR_HIP;                                                                           \
        }                                                                                                 \
    } while (0)
        PEA_HIP_C2(hipMalloc((void **)&hub_dev, (size_t)nh * sizeof(int)));
        PEA_HIP_C2(hipMalloc((void **)&bounds_dev, bounds.size() * sizeof(int)));
        PEA_HIP_C2(hipMemcpyAsync(hub_dev, sliced_rows.data(), (size_t)nh * sizeof(int), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(slice_bounds, dim3((unsigned)((bounds.size() + 255) / 256)), dim3(256), 0, stream, nh, S, E, hub_dev,
                           k64_out, bounds_dev);
        PEA_HIP_C2(hipGetLastError());
        PEA_HIP_C2(hipMemcpyAsync(bounds.data(), bounds_dev, bounds.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
        PEA_HIP_C2(hipStreamSynchronize(stream));
#undef PEA_HIP_C2
        std::vector<std::vector<LongItem>> per_slice((size_t)S);
        for (int h = 0; h < nh; ++h) {
            hub_rows.push_back(sliced_rows[(size_t)h]);
            hub_first.push_back(slots);
            int count = 0;
            for (int sl = 0; sl < S; ++sl) {
                const int b0 = bounds[(size_t)h * (S + 1) + sl], e0 = bounds[(size_t)h * (S + 1) + sl + 1];
                if (e0 <= b0) continue;
                const int nch = (e0 - b0 + chunk - 1) / chunk, len = (e0 - b0 + nch - 1) / nch;
                for (int c = 0; c < nch; ++c) {
                    const int b = b0 + c * len;
                    per_slice[(size_t)sl].push_back({sliced_rows[(size_t)h], b, std::min(b + len, e0), slots++});
                    ++count;
                }
            }
            hub_count.push_back(count);
        }
        std::vector<LongItem> head;
        const LongItem pad = {0, 0, 0, -2};
        for (int ph = 0; ph < S / 8; ++ph) {
            size_t rounds = 0;
            for (int x = 0; x < 8; ++x) rounds = std::max(rounds, (per_slice[(size_t)ph * 8 + x].size() + 3) / 4);
            for (size_t q = 0; q < rounds; ++q)
                for (int x = 0; x < 8; ++x) {
                    const std::vector<LongItem> &v = per_slice[(size_t)ph * 8 + x];
                    for (size_t w = 0; w < 4; ++w) head.push_back(q * 4 + w < v.size() ? v[q * 4 + w] : pad);
                }
        }
        head.insert(head.end(), items.begin(), items.end());
        items.swap(head);
    }
    cleanup();
    R.max_deg = max_deg;
    R.edges_long = R.edges_owned - R.edges_short;
    R.n_short0 = (int)zero_rows.size();
    zero_rows.insert(zero_rows.end(), short_rows.begin(), short_rows.end());  // edge-less rows first
    short_rows.swap(zero_rows);
    R.n_short = (int)short_rows.size();
    {
        std::vector<unsigned char> flag((size_t)N);
        for (int64_t i = 0; i < N; ++i) flag[(size_t)i] = rp[(size_t)i + 1] == rp[(size_t)i];
        PEA_TRY(upload(flag, &R.deg0));
    }
    R.n_long = (int)items.size();
    R.n_direct = 0;
    for (const LongItem &it : items) R.n_direct += it.slot == -1;
    R.n_hub = (int)hub_rows.size();
    R.n_slots = slots;
    plan->max_slots = std::max(plan->max_slots, slots);
    PEA_TRY(upload(short_rows, &R.short_rows));
    PEA_TRY(upload(items, &R.long_items));
    PEA_TRY(upload(hub_rows, &R.hub_rows));
    PEA_TRY(upload(hub_first, &R.hub_first));
    PEA_TRY(upload(hub_count, &R.hub_count));
    (void)rc;
    return PEA_OK;
#undef PEA_HIP_C
}

__global__ void rename_sources(int64_t E, const int *__restrict__ col, const int *__restrict__ slot_of_node,
                               int *__restrict__ col_slot, int *err) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int s = slot_of_node[col[e]];
    if (s < 0) atomicOr(err, 1);
    col_slot[e] = s < 0 ? 0 : s;
}

__global__ void scatter_to_slots(int64_t N, const int *__restrict__ slot_of_node, const float *__restrict__ src,
                                 float *__restrict__ dst) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int s = slot_of_node[v];
    if (s >= 0) dst[s] = src[v];
}

__global__ void count_ids(int64_t E, const int *__restrict__ col, int *__restrict__ cnt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(&cnt[col[e]], 1);
}

__global__ void encode_hot(int64_t E, const int *__restrict__ col, const int *__restrict__ rank_of, int *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int j = col[e], r = rank_of[j];
    out[e] = r >= 0 ? -(r + 2) : j;
}

void free_relation(Relation &R) {
    for (HotVariant &h : R.hot) {
        (void)hipFree(h.col);
        (void)hipFree(h.nodes);
    }
    (void)hipFree(R.col_slot); (void)hipFree(R.dinv_row_slot); (void)hipFree(R.dinv_col_slot);
    (void)hipFree(R.slot_of_node); (void)hipFree(R.need_rows);
    (void)hipFree(R.rowptr); (void)hipFree(R.col); (void)hipFree(R.dinv_row); (void)hipFree(R.dinv_col); (void)hipFree(R.invdeg); (void)hipFree(R.eid);
    (void)hipFree(R.short_rows); (void)hipFree(R.long_items); (void)hipFree(R.deg0);
    (void)hipFree(R.hub_rows); (void)hipFree(R.hub_first); (void)hipFree(R.hub_count);
}

}  // namespace

int ensure_dinv(pea_plan *plan, int rel, bool from_col, hipStream_t stream) {
    Relation &R = plan->rels[(size_t)rel];
    float **slot = from_col ? &R.dinv_col : &R.dinv_row;
    if (*slot) return PEA_OK;
    const int64_t N = plan->N;
    int *cnt = nullptr;
    PEA_HIP(hipMalloc((void **)slot, (size_t)N * sizeof(float)));
    PEA_HIP(hipMalloc((void **)&cnt, (size_t)N * sizeof(int)));
    PEA_HIP(hipMemsetAsync(cnt, 0, (size_t)N * sizeof(int), stream));
    if (!from_col && R.e_kept > 0) {
        hipLaunchKernelGGL(count_sources, dim3((unsigned)((R.e_kept + 255) / 256)), dim3(256), 0, stream, R.e_kept,
                           R.col, cnt);
    }
    hipLaunchKernelGGL(dinv_from_counts, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, cnt, R.rowptr,
                       from_col ? 1 : 0, (plan->flags & PEA_PLAN_SELF_LOOPS) ? 1 : 0, *slot);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(cnt);
    PEA_REQUIRE(e == hipSuccess, PEA_ERR_HIP, "gcn norm: %s", hipGetErrorString(e));
    return PEA_OK;
}

int ensure_hot(pea_plan *plan, int rel, int K, bool via_slots, hipStream_t stream, const HotVariant **out) {
    *out = nullptr;
    Relation &R = plan->rels[(size_t)rel];
    const char *env = getenv("PEA_HOT_MIN_EDGES");
    const int64_t min_edges = env ? atoll(env) : kHotMinEdges;
    if (K < 8 || R.e_kept < min_edges) return PEA_OK;
    for (const HotVariant &h : R.hot)
        if (h.K == K && h.via_slots == via_slots) {
            if (h.col) *out = &h;
            return PEA_OK;
        }
    const int *col = via_slots ? R.col_slot : R.col;
    PEA_REQUIRE(col != nullptr, PEA_ERR_ARG, "relation %d has no exchange layout (pea_plan_set_sources)", rel);
    const int64_t n_ids = via_slots ? std::max<int64_t>(R.slots_per_rank * plan->shard_world, 1) : plan->N;
    HotVariant hv;
    hv.K = K;
    hv.via_slots = via_slots;
    int *cnt = nullptr;
    PEA_HIP(hipMalloc((void **)&cnt, (size_t)n_ids * sizeof(int)));
    std::vector<int> h_cnt((size_t)n_ids);
    hipError_t e = hipMemsetAsync(cnt, 0, (size_t)n_ids * sizeof(int), stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(count_ids, dim3((unsigned)((R.e_kept + 255) / 256)), dim3(256), 0, stream, R.e_kept, col, cnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_cnt.data(), cnt, (size_t)n_ids * sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        (void)hipFree(cnt);
        PEA_REQUIRE(false, PEA_ERR_HIP, "hot sources: %s", hipGetErrorString(e));
    }
    // the K most frequent sources (ties: smaller id first, so the choice is deterministic)
    std::vector<int> ids((size_t)n_ids);
    for (int64_t i = 0; i < n_ids; ++i) ids[(size_t)i] = (int)i;
    const size_t k = (size_t)std::min<int64_t>(K, n_ids);
    std::partial_sort(ids.begin(), ids.begin() + (ptrdiff_t)k, ids.end(), [&](int a, int b) {
        return h_cnt[(size_t)a] != h_cnt[(size_t)b] ? h_cnt[(size_t)a] > h_cnt[(size_t)b] : a < b;
    });
    std::vector<int> nodes((size_t)K, ids[0]), rank_of((size_t)n_ids, -1);
    for (size_t r = 0; r < k; ++r) {
        if (h_cnt[(size_t)ids[r]] == 0) break;  // fewer than K distinct sources
        nodes[r] = ids[r];
        rank_of[(size_t)ids[r]] = (int)r;
        hv.hot_edges += h_cnt[(size_t)ids[r]];
    }
    const char *envf = getenv("PEA_HOT_MIN_FRACTION");
    const double min_frac = envf ? atof(envf) : kHotMinFraction;
    if ((double)hv.hot_edges >= min_frac * (double)R.e_kept) {
        int *rank_dev = nullptr;
        e = hipMalloc((void **)&rank_dev, (size_t)n_ids * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&hv.col, (size_t)R.e_kept * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&hv.nodes, (size_t)K * sizeof(int));
        if (e == hipSuccess) e = hipMemcpyAsync(rank_dev, rank_of.data(), (size_t)n_ids * sizeof(int), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hv.nodes, nodes.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(encode_hot, dim3((unsigned)((R.e_kept + 255) / 256)), dim3(256), 0, stream, R.e_kept, col, rank_dev, hv.col);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(rank_dev);
        if (e != hipSuccess) {
            (void)hipFree(hv.col);
            (void)hipFree(hv.nodes);
            (void)hipFree(cnt);
            PEA_REQUIRE(false, PEA_ERR_HIP, "hot sources: %s", hipGetErrorString(e));
        }
    }
    (void)hipFree(cnt);
    R.hot.reserve(8);  // pointers into the vector are handed out: never reallocate (a handful of variants at most)
    PEA_REQUIRE(R.hot.size() < 8, PEA_ERR_ARG, "relation %d: too many hot-source variants", rel);
    R.hot.push_back(hv);
    if (hv.col) *out = &R.hot.back();
    return PEA_OK;
}

int ensure_dinv_slots(pea_plan *plan, int rel, bool from_col, hipStream_t stream) {
    PEA_TRY(ensure_dinv(plan, rel, from_col, stream));
    Relation &R = plan->rels[(size_t)rel];
    float **slot = from_col ? &R.dinv_col_slot : &R.dinv_row_slot;
    if (*slot) return PEA_OK;
    PEA_REQUIRE(R.slot_of_node != nullptr, PEA_ERR_ARG, "relation %d has no exchange layout (pea_plan_set_sources)", rel);
    const size_t n = (size_t)std::max<int64_t>(R.slots_per_rank * plan->shard_world, 1);
    PEA_HIP(hipMalloc((void **)slot, n * sizeof(float)));
    PEA_HIP(hipMemsetAsync(*slot, 0, n * sizeof(float), stream));
    hipLaunchKernelGGL(scatter_to_slots, dim3((unsigned)((plan->N + 255) / 256)), dim3(256), 0, stream, plan->N,
                       R.slot_of_node, from_col ? R.dinv_col : R.dinv_row, *slot);
    PEA_HIP(hipGetLastError());
    PEA_HIP(hipStreamSynchronize(stream));
    return PEA_OK;
}

}  // namespace pea

extern "C" int pea_plan_set_owned_rows(pea_plan *plan, const int32_t *rows, int64_t n, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(plan && n >= 0 && (n == 0 || rows), PEA_ERR_ARG, "set_owned_rows: bad argument");
    (void)hipFree(plan->owned_rows);
    plan->owned_rows = nullptr;
    plan->n_owned = n;
    if (n > 0) {
        PEA_HIP(hipMalloc((void **)&plan->owned_rows, (size_t)n * sizeof(int)));
        PEA_HIP(hipMemcpyAsync(plan->owned_rows, rows, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, stream));
        PEA_HIP(hipStreamSynchronize(stream));
    }
    return PEA_OK;
}

extern "C" int pea_plan_set_owned_split(pea_plan *plan, int64_t n_first) {
    PEA_REQUIRE(plan && n_first >= 0 && n_first <= plan->n_owned, PEA_ERR_ARG, "set_owned_split: %lld of %lld owned rows",
                (long long)n_first, plan ? (long long)plan->n_owned : 0LL);
    plan->n_owned_first = n_first;
    return PEA_OK;
}

extern "C" int pea_plan_set_sources(pea_plan *plan, int relation, const int32_t *slot_of_node, int64_t slots_per_rank,
                                    const int32_t *need_rows, int64_t n_need, void *stream_) {
    using namespace pea;
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(plan && relation >= 0 && relation < (int)plan->rels.size() && slot_of_node && slots_per_rank >= 0 &&
                    n_need >= 0 && (n_need == 0 || need_rows), PEA_ERR_ARG, "set_sources: bad argument");
    PEA_REQUIRE(slots_per_rank * plan->shard_world < (int64_t)INT32_MAX, PEA_ERR_ARG, "set_sources: too many slots");
    Relation &R = plan->rels[(size_t)relation];
    (void)hipFree(R.col_slot); (void)hipFree(R.slot_of_node); (void)hipFree(R.need_rows);
    (void)hipFree(R.dinv_row_slot); (void)hipFree(R.dinv_col_slot);
    R.col_slot = R.slot_of_node = R.need_rows = nullptr;
    R.dinv_row_slot = R.dinv_col_slot = nullptr;
    R.slots_per_rank = slots_per_rank;
    R.n_need = n_need;
    PEA_HIP(hipMalloc((void **)&R.slot_of_node, (size_t)plan->N * sizeof(int)));
    PEA_HIP(hipMemcpyAsync(R.slot_of_node, slot_of_node, (size_t)plan->N * sizeof(int), hipMemcpyDeviceToDevice, stream));
    if (n_need > 0) {
        PEA_HIP(hipMalloc((void **)&R.need_rows, (size_t)n_need * sizeof(int)));
        PEA_HIP(hipMemcpyAsync(R.need_rows, need_rows, (size_t)n_need * sizeof(int), hipMemcpyDeviceToDevice, stream));
    }
    PEA_HIP(hipMalloc((void **)&R.col_slot, (size_t)std::max<int64_t>(R.e_kept, 1) * sizeof(int)));
    int *err = nullptr, herr = 0;
    PEA_HIP(hipMalloc((void **)&err, sizeof(int)));
    PEA_HIP(hipMemsetAsync(err, 0, sizeof(int), stream));
    if (R.e_kept > 0)
        hipLaunchKernelGGL(rename_sources, dim3((unsigned)((R.e_kept + 255) / 256)), dim3(256), 0, stream, R.e_kept, R.col,
                           R.slot_of_node, R.col_slot, err);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, stream);
    // content hash of the row list: relations handed the SAME list share one first-layer transform job (model.hip)
    std::vector<int> host_rows((size_t)n_need);
    if (e == hipSuccess && n_need > 0)
        e = hipMemcpyAsync(host_rows.data(), R.need_rows, (size_t)n_need * sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    unsigned long long hsh = 1469598103934665603ull;  // FNV-1a
    for (int v : host_rows) hsh = (hsh ^ (unsigned long long)(unsigned)v) * 1099511628211ull;
    R.need_hash = hsh;
    (void)hipFree(err);
    PEA_REQUIRE(e == hipSuccess, PEA_ERR_HIP, "set_sources: %s", hipGetErrorString(e));
    PEA_REQUIRE(herr == 0, PEA_ERR_ARG, "set_sources: a source node of relation %d has no slot", relation);
    return PEA_OK;
}

extern "C" int pea_plan_create(int64_t num_nodes, int n_relations, const int64_t *const *coo_host,
                               const int64_t *num_edges_host, int flags, int gather_row_bytes, int shard_rank,
                               int shard_world, int shard_tile, void *stream_, pea_plan **out) {
    using namespace pea;
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(out != nullptr, PEA_ERR_ARG, "plan: out is null");
    *out = nullptr;
    PEA_REQUIRE(num_nodes > 0 && num_nodes < (int64_t)INT32_MAX - 64, PEA_ERR_ARG, "plan: num_nodes=%lld", (long long)num_nodes);
    PEA_REQUIRE(n_relations > 0 && coo_host && num_edges_host, PEA_ERR_ARG, "plan: no relations");
    PEA_REQUIRE(shard_world >= 1 && shard_rank >= 0 && shard_rank < shard_world && shard_tile > 0, PEA_ERR_ARG,
                "plan: bad shard (rank %d of %d, tile %d)", shard_rank, shard_world, shard_tile);
    PEA_REQUIRE(pea_device_count() > 0, PEA_ERR_DEVICE, "no gfx950 device visible");
    pea_plan *plan = new pea_plan();
    plan->N = num_nodes;
    plan->flags = flags;
    plan->gather_row_bytes = gather_row_bytes > 0 ? gather_row_bytes : 0;
    plan->shard_rank = shard_rank;
    plan->shard_world = shard_world;
    plan->shard_tile = shard_tile;
    plan->rels.resize((size_t)n_relations);
    for (int r = 0; r < n_relations; ++r) {
        int rc = PEA_ERR_ARG;
        if (num_edges_host[r] > 0 && coo_host[r] == nullptr) set_error("plan: relation %d has a null edge_index", r);
        else rc = build_relation(plan, plan->rels[(size_t)r], coo_host[r], num_edges_host[r], stream);
        if (rc != PEA_OK) {
            pea_plan_destroy(plan);
            return rc;
        }
    }
    *out = plan;
    return PEA_OK;
}

extern "C" int pea_plan_destroy(pea_plan *plan) {
    if (!plan) return PEA_OK;
    for (auto &R : plan->rels) pea::free_relation(R);
    (void)hipFree(plan->owned_rows);
    (void)hipFree(plan->ones);
    delete plan;
    return PEA_OK;
}

extern "C" int pea_plan_relation_info(const pea_plan *plan, int relation, int64_t *info) {
    PEA_REQUIRE(plan && info && relation >= 0 && relation < (int)plan->rels.size(), PEA_ERR_ARG, "plan info: bad argument");
    const pea::Relation &R = plan->rels[(size_t)relation];
    info[0] = R.e_kept; info[1] = R.max_deg; info[2] = R.n_short; info[3] = R.n_long;
    info[4] = R.n_hub; info[5] = R.n_slots; info[6] = R.rows_owned; info[7] = R.edges_owned;
    info[8] = R.slices;
    return PEA_OK;
}

extern "C" int pea_plan_export_csr(const pea_plan *plan, int relation, int32_t *rowptr, int32_t *col, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    PEA_REQUIRE(plan && relation >= 0 && relation < (int)plan->rels.size(), PEA_ERR_ARG, "export csr: bad argument");
    const pea::Relation &R = plan->rels[(size_t)relation];
    if (rowptr) PEA_HIP(hipMemcpyAsync(rowptr, R.rowptr, (size_t)(plan->N + 1) * sizeof(int), hipMemcpyDeviceToDevice, stream));
    if (col && R.e_kept > 0) PEA_HIP(hipMemcpyAsync(col, R.col, (size_t)R.e_kept * sizeof(int), hipMemcpyDeviceToDevice, stream));
    return PEA_OK;
}
