// Device-side BPR negative sampler for gfx950.  An ADDITION to the bit-exact host path
// (graph_recsys_benchmark_amd/utils/sampling.py mirrors the reference's numpy / random streams id for id,
// reference datasets/movielens.py:920-940): same two strategies, same output layout, its own counter-based stream.
//   'random':  i- uniform over the item block            (movielens.py:925-928: np.random.randint)
//   'unseen':  i- uniform over the items the user has no TRAINING interaction with
//              (movielens.py:929-937: choices(test positives + never-seen items) -- that pool is exactly
//              all items minus the user's training positives), by rejection against a sorted key table.
// Stream: Philox4x32-10 (Salmon et al. 2011), key = seed, counter = (row index lo, hi, attempt, offset); the first
// 32-bit output word w maps to an item as floor(w * num_items / 2^32).  oracle/philox.py restates it in numpy.
#include "common.h"

namespace pea {
namespace {

struct U4 {
    unsigned x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
        c = U4{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// first index with keys[idx] >= key
__device__ __forceinline__ bool contains(const long long *__restrict__ keys, long long n, long long key) {
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo < n && keys[lo] == key;
}

constexpr int kMaxAttempts = 64;  // a user who has seen almost every item: the last draw is kept and flagged

__global__ __launch_bounds__(256) void sample_kernel(long long n_pos, int k, const long long *__restrict__ pos_u,
                                                     const long long *__restrict__ pos_i, long long item_lo,
                                                     long long num_items, const long long *__restrict__ keys, long long n_keys,
                                                     unsigned seed_lo, unsigned seed_hi, unsigned offset,
                                                     long long *__restrict__ out, long long ld_out, int *__restrict__ exhausted) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_pos * k) return;
    const long long p = r / k;
    const long long u = pos_u[p];
    long long neg = item_lo;
    bool ok = false;
    for (int attempt = 0; attempt < kMaxAttempts && !ok; ++attempt) {
        const U4 w = philox4x32_10(U4{(unsigned)r, (unsigned)((unsigned long long)r >> 32), (unsigned)attempt, offset}, seed_lo, seed_hi);
        const long long j = (long long)(((unsigned long long)w.x * (unsigned long long)num_items) >> 32);
        neg = item_lo + j;
        ok = keys == nullptr || !contains(keys, n_keys, u * num_items + j);
    }
    if (!ok) atomicAdd(exhausted, 1);
    out[r * ld_out + 0] = u;
    out[r * ld_out + 1] = pos_i[p];
    out[r * ld_out + 2] = neg;
}

}  // namespace
}  // namespace pea

using namespace pea;

extern "C" int pea_sample_negatives(int64_t n_pos, int k, const int64_t *pos_u, const int64_t *pos_i, int64_t item_lo,
                                    int64_t num_items, const int64_t *seen_keys_sorted, int64_t n_keys, uint64_t seed,
                                    uint32_t offset, int64_t *out_triples, int64_t ld_out, int *exhausted, void *stream) {
    PEA_REQUIRE(n_pos >= 0 && k > 0 && num_items > 0 && num_items < (1ll << 32) && ld_out >= 3, PEA_ERR_ARG,
                "sample_negatives: bad sizes (n_pos %lld, k %d, num_items %lld, ld_out %lld)", (long long)n_pos, k,
                (long long)num_items, (long long)ld_out);
    PEA_REQUIRE((pos_u && pos_i && out_triples && exhausted) || n_pos == 0, PEA_ERR_ARG, "sample_negatives: null argument");
    PEA_REQUIRE(n_keys >= 0 && (seen_keys_sorted || n_keys == 0), PEA_ERR_ARG, "sample_negatives: key table missing");
    if (n_pos == 0) return PEA_OK;
    const long long total = (long long)n_pos * k;
    hipStream_t st = (hipStream_t)stream;
    PEA_MEMSET_ASYNC(exhausted, 0, sizeof(int), st);
    ProfScope ps("sample_negatives", st);
    PEA_LAUNCH(sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (long long)n_pos, k,
                       reinterpret_cast<const long long *>(pos_u), reinterpret_cast<const long long *>(pos_i),
                       (long long)item_lo, (long long)num_items, reinterpret_cast<const long long *>(seen_keys_sorted),
                       (long long)n_keys, (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), (unsigned)offset,
                       reinterpret_cast<long long *>(out_triples), (long long)ld_out, exhausted);
    PEA_HIP(hipGetLastError());
    return PEA_OK;
}
