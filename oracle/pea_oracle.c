/*
 * ORACLE / TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (fp32, int64 ids) CPU restatement of the reference's PEA metapath
 * aggregation path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library -- as the checker / the reported CPU
 * baseline, never as part of the product path (graph_recsys_benchmark_amd/
 * never imports anything under oracle/).
 *
 * What each function follows (paths relative to /root/reference):
 *   orc_gat_conv   torch_geometric 1.5.0 GATConv  (ctor sites graph_recsys_benchmark/models/peagat.py:16-21,
 *                  call site models/base.py:138-139; algorithm SURVEY.md Appendix A.1)
 *   orc_gcn_conv   torch_geometric 1.5.0 GCNConv  (models/peagcn.py:16-21; Appendix A.2)
 *   orc_sage_conv  torch_geometric 1.5.0 SAGEConv (models/peasage.py:16-21; Appendix A.3)
 *   orc_relu       F.relu between steps            (models/base.py:138)
 *   orc_fuse       channel stack + ablation mask + 'att' / 'mean' fusion (models/base.py:193-203)
 *   orc_predict    PEABaseRecsysModel.predict      (models/base.py:208-214)
 *   orc_bpr_loss   GraphRecsysModel.loss cf term   (models/base.py:46-48)
 *   orc_entity_reg entity-aware regulariser        (models/base.py:50-73)
 *
 * torch-geometric 1.5.0 + torch-scatter 2.0.5 (requirements.txt:41,43) are not
 * vendored in the reference tree and not installed here, so the conv arithmetic
 * is restated from that release's published algorithm: PARITY OF THE CONV
 * ARITHMETIC IS UNPINNED.  The reference-owned parts (fusion, predict, loss) are
 * pinned by the .npz fixtures in tests/golden, produced by running the reference's own
 * models/base.py (oracle/make_golden.py).
 *
 * The op sequence mirrors MessagePassing.propagate(): materialised gather
 * ([M,F] temporaries), elementwise message, then a scatter that adds messages
 * in EDGE ORDER (torch-scatter's CPU kernel is one sequential loop).  Loops that
 * torch runs in parallel (gather, elementwise, GEMM rows) are OpenMP-parallel;
 * the scatters are sequential like the reference's.
 *
 * Layout: edge_index is int64 [2,E] row-major (ei[0..E) = source j, ei[E..2E) = target i).
 * Every function returns 0 on success, negative on bad arguments / OOM.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* y[n, out] = x[n, in] . W[out, in]^T (+ b)   (torch.nn.Linear layout) */
static void linear_oi(int64_t n, int in, int out, const float *x, int64_t ldx, const float *W,
                      const float *b, float *y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        const float *xr = x + r * ldx;
        float *yr = y + r * (int64_t)out;
        for (int o = 0; o < out; ++o) {
            const float *w = W + (int64_t)o * in;
            float acc = 0.f;
            for (int k = 0; k < in; ++k) acc += xr[k] * w[k];
            yr[o] = b ? acc + b[o] : acc;
        }
    }
}

/* y[n, out] = x[n, in] . W[in, out]   (GCNConv.weight layout) */
static void linear_io(int64_t n, int in, int out, const float *x, const float *W, float *y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        const float *xr = x + r * (int64_t)in;
        float *yr = y + r * (int64_t)out;
        for (int o = 0; o < out; ++o) yr[o] = 0.f;
        for (int k = 0; k < in; ++k) {
            const float xv = xr[k];
            const float *w = W + (int64_t)k * out;
            for (int o = 0; o < out; ++o) yr[o] += xv * w[o];
        }
    }
}

/* remove self loops, append one loop per node; returns M and fills row/col (size E+N) */
static int64_t rewrite_self_loops(int64_t N, int64_t E, const int64_t *ei, int64_t *row, int64_t *col) {
    int64_t m = 0;
    for (int64_t e = 0; e < E; ++e) {
        const int64_t j = ei[e], i = ei[E + e];
        if (j != i) { row[m] = j; col[m] = i; ++m; }
    }
    for (int64_t v = 0; v < N; ++v) { row[m] = v; col[m] = v; ++m; }
    return m;
}

static int check_ids(int64_t N, int64_t E, const int64_t *ei) {
    for (int64_t e = 0; e < 2 * E; ++e)
        if (ei[e] < 0 || ei[e] >= N) return -2;
    return 0;
}

int orc_gat_conv(int64_t N, int64_t E, const int64_t *ei, int Fin, int heads, int Fout,
                 const float *x, const float *W, const float *att_i, const float *att_j,
                 const float *bias, float neg_slope, int concat, float *out) {
    if (N <= 0 || E < 0 || Fin <= 0 || heads <= 0 || Fout <= 0) return -1;
    if (check_ids(N, E, ei)) return -2;
    const int HF = heads * Fout;
    float *h = (float *)malloc(sizeof(float) * N * HF);
    int64_t *row = (int64_t *)malloc(sizeof(int64_t) * (E + N));
    int64_t *col = (int64_t *)malloc(sizeof(int64_t) * (E + N));
    if (!h || !row || !col) { free(h); free(row); free(col); return -3; }
    linear_oi(N, Fin, HF, x, Fin, W, NULL, h);
    const int64_t M = rewrite_self_loops(N, E, ei, row, col);

    /* propagate(): x_j = h[row], x_i = h[col] materialised */
    float *xj = (float *)malloc(sizeof(float) * M * HF);
    float *xi = (float *)malloc(sizeof(float) * M * HF);
    float *alpha = (float *)malloc(sizeof(float) * M * heads);
    float *amax = (float *)malloc(sizeof(float) * N * heads);
    float *asum = (float *)malloc(sizeof(float) * N * heads);
    float *acc = (float *)calloc((size_t)N * HF, sizeof(float));
    int rc = 0;
    if (!xj || !xi || !alpha || !amax || !asum || !acc) { rc = -3; goto done; }
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < M; ++e) {
        memcpy(xj + e * HF, h + row[e] * HF, sizeof(float) * HF);
        memcpy(xi + e * HF, h + col[e] * HF, sizeof(float) * HF);
    }
    /* message(): alpha = leaky_relu((x_i*att_i).sum(-1) + (x_j*att_j).sum(-1)) */
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < M; ++e)
        for (int k = 0; k < heads; ++k) {
            float si = 0.f, sj = 0.f;
            for (int f = 0; f < Fout; ++f) {
                si += xi[e * HF + k * Fout + f] * att_i[k * Fout + f];
                sj += xj[e * HF + k * Fout + f] * att_j[k * Fout + f];
            }
            float a = si + sj;
            alpha[e * heads + k] = a > 0.f ? a : a * neg_slope;
        }
    /* softmax(alpha, col, N): scatter_max, exp, scatter_add, divide (+1e-16) */
    for (int64_t v = 0; v < N * heads; ++v) { amax[v] = -INFINITY; asum[v] = 0.f; }
    for (int64_t e = 0; e < M; ++e)
        for (int k = 0; k < heads; ++k) {
            float *m = amax + col[e] * heads + k;
            if (alpha[e * heads + k] > *m) *m = alpha[e * heads + k];
        }
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < M; ++e)
        for (int k = 0; k < heads; ++k)
            alpha[e * heads + k] = expf(alpha[e * heads + k] - amax[col[e] * heads + k]);
    for (int64_t e = 0; e < M; ++e)
        for (int k = 0; k < heads; ++k) asum[col[e] * heads + k] += alpha[e * heads + k];
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < M; ++e)
        for (int k = 0; k < heads; ++k)
            alpha[e * heads + k] = alpha[e * heads + k] / (asum[col[e] * heads + k] + 1e-16f);
    /* msg = x_j * alpha (in place), then scatter add in edge order */
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < M; ++e)
        for (int k = 0; k < heads; ++k)
            for (int f = 0; f < Fout; ++f) xj[e * HF + k * Fout + f] *= alpha[e * heads + k];
    for (int64_t e = 0; e < M; ++e) {
        float *o = acc + col[e] * HF;
        const float *m = xj + e * HF;
        for (int c = 0; c < HF; ++c) o[c] += m[c];
    }
    if (concat) {
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < N; ++v)
            for (int c = 0; c < HF; ++c) out[v * HF + c] = bias ? acc[v * HF + c] + bias[c] : acc[v * HF + c];
    } else {
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < N; ++v)
            for (int f = 0; f < Fout; ++f) {
                float s = 0.f;
                for (int k = 0; k < heads; ++k) s += acc[v * HF + k * Fout + f];
                s = s / (float)heads;
                out[v * Fout + f] = bias ? s + bias[f] : s;
            }
    }
done:
    free(h); free(row); free(col); free(xj); free(xi); free(alpha); free(amax); free(asum); free(acc);
    return rc;
}

int orc_gcn_conv(int64_t N, int64_t E, const int64_t *ei, int Fin, int Fout, const float *x,
                 const float *W, const float *bias, int deg_from_col, float *out) {
    if (N <= 0 || E < 0 || Fin <= 0 || Fout <= 0) return -1;
    if (check_ids(N, E, ei)) return -2;
    float *h = (float *)malloc(sizeof(float) * N * Fout);
    int64_t *row = (int64_t *)malloc(sizeof(int64_t) * (E + N));
    int64_t *col = (int64_t *)malloc(sizeof(int64_t) * (E + N));
    float *deg = (float *)calloc((size_t)N, sizeof(float));
    float *acc = (float *)calloc((size_t)N * Fout, sizeof(float));
    float *msg = NULL, *norm = NULL;
    int rc = 0;
    if (!h || !row || !col || !deg || !acc) { rc = -3; goto done; }
    linear_io(N, Fin, Fout, x, W, h);
    const int64_t M = rewrite_self_loops(N, E, ei, row, col);
    msg = (float *)malloc(sizeof(float) * M * Fout);
    norm = (float *)malloc(sizeof(float) * M);
    if (!msg || !norm) { rc = -3; goto done; }
    /* deg = scatter_add(ones, row)  (PyG <= 1.5.0; 'col' = PyG >= 1.6 gcn_norm) */
    for (int64_t e = 0; e < M; ++e) deg[deg_from_col ? col[e] : row[e]] += 1.0f;
    for (int64_t v = 0; v < N; ++v) {
        float d = powf(deg[v], -0.5f);
        deg[v] = isinf(d) ? 0.f : d;
    }
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < M; ++e) {
        norm[e] = deg[row[e]] * 1.0f * deg[col[e]];
        const float *hj = h + row[e] * Fout;
        for (int c = 0; c < Fout; ++c) msg[e * Fout + c] = norm[e] * hj[c];
    }
    for (int64_t e = 0; e < M; ++e) {
        float *o = acc + col[e] * Fout;
        for (int c = 0; c < Fout; ++c) o[c] += msg[e * Fout + c];
    }
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N; ++v)
        for (int c = 0; c < Fout; ++c) out[v * Fout + c] = bias ? acc[v * Fout + c] + bias[c] : acc[v * Fout + c];
done:
    free(h); free(row); free(col); free(deg); free(acc); free(msg); free(norm);
    return rc;
}

int orc_sage_conv(int64_t N, int64_t E, const int64_t *ei, int Fin, int Fout, const float *x,
                  const float *Wrel, const float *brel, const float *Wroot, float *out) {
    if (N <= 0 || E < 0 || Fin <= 0 || Fout <= 0) return -1;
    if (check_ids(N, E, ei)) return -2;
    float *msg = (float *)malloc(sizeof(float) * (E > 0 ? E : 1) * Fin);
    float *sum = (float *)calloc((size_t)N * Fin, sizeof(float));
    float *cnt = (float *)calloc((size_t)N, sizeof(float));
    float *rel = (float *)malloc(sizeof(float) * N * Fout);
    float *root = (float *)malloc(sizeof(float) * N * Fout);
    int rc = 0;
    if (!msg || !sum || !cnt || !rel || !root) { rc = -3; goto done; }
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) memcpy(msg + e * Fin, x + ei[e] * Fin, sizeof(float) * Fin);
    for (int64_t e = 0; e < E; ++e) {
        float *o = sum + ei[E + e] * Fin;
        for (int c = 0; c < Fin; ++c) o[c] += msg[e * Fin + c];
        cnt[ei[E + e]] += 1.0f;
    }
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N; ++v) {
        const float d = cnt[v] < 1.0f ? 1.0f : cnt[v];
        for (int c = 0; c < Fin; ++c) sum[v * Fin + c] = sum[v * Fin + c] / d;
    }
    linear_oi(N, Fin, Fout, sum, Fin, Wrel, brel, rel);
    linear_oi(N, Fin, Fout, x, Fin, Wroot, NULL, root);
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N * Fout; ++v) out[v] = rel[v] + root[v];
done:
    free(msg); free(sum); free(cnt); free(rel); free(root);
    return rc;
}

void orc_relu(int64_t n, float *x) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f;
}

/* X: [N, P, R] stacked channel outputs (torch.cat(dim=1)); att: [P, R] or NULL for 'mean';
 * masked_p >= 0 zeroes that channel BEFORE fusion (it keeps logit 0 in the softmax). */
int orc_fuse(int64_t N, int P, int R, const float *X, const float *att, int masked_p, float *out) {
    if (N <= 0 || P <= 0 || P > 64 || R <= 0 || masked_p >= P) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float score[64];
        const float *xn = X + n * (int64_t)P * R;
        float *o = out + n * (int64_t)R;
        if (!att) {
            for (int r = 0; r < R; ++r) {
                float s = 0.f;
                for (int p = 0; p < P; ++p) s += (p == masked_p) ? 0.f : xn[p * R + r];
                o[r] = s / (float)P;
            }
            continue;
        }
        float mx = -INFINITY;
        for (int p = 0; p < P; ++p) {
            float s = 0.f;
            if (p != masked_p)
                for (int r = 0; r < R; ++r) s += xn[p * R + r] * att[p * R + r];
            score[p] = s;
            if (s > mx) mx = s;
        }
        float den = 0.f;
        for (int p = 0; p < P; ++p) { score[p] = expf(score[p] - mx); den += score[p]; }
        for (int p = 0; p < P; ++p) score[p] = score[p] / den;
        for (int r = 0; r < R; ++r) {
            float s = 0.f;
            for (int p = 0; p < P; ++p) s += (p == masked_p) ? 0.f : xn[p * R + r] * score[p];
            o[r] = s;
        }
    }
    return 0;
}

/* pred[b] = fc2(relu(fc1([repr[u_b] || repr[i_b]]))) ; fc1_w [R,2R], fc2_w [1,R] */
int orc_predict(int64_t B, int R, int64_t N, const float *repr, const int64_t *unids, int64_t ustride,
                const int64_t *inids, int64_t istride, const float *fc1_w, const float *fc1_b,
                const float *fc2_w, const float *fc2_b, float *pred) {
    if (B < 0 || R <= 0 || R > 1024) return -1;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t u = unids[b * ustride], i = inids[b * istride];
        if (u < 0 || u >= N || i < 0 || i >= N) return -2;
    }
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float *ur = repr + unids[b * ustride] * R;
        const float *ir = repr + inids[b * istride] * R;
        float o = 0.f;
        for (int k = 0; k < R; ++k) {
            const float *w = fc1_w + (int64_t)k * 2 * R;
            float a = 0.f;
            for (int c = 0; c < R; ++c) a += ur[c] * w[c];
            for (int c = 0; c < R; ++c) a += ir[c] * w[R + c];
            a += fc1_b[k];
            a = a > 0.f ? a : 0.f;
            o += a * fc2_w[k];
        }
        pred[b] = o + fc2_b[0];
    }
    return 0;
}

/* -sum(log(sigmoid(pos - neg)))  -- no clamp, may return +inf like the reference */
float orc_bpr_loss(int64_t B, const float *pos, const float *neg) {
    float s = 0.f;
    for (int64_t b = 0; b < B; ++b) {
        const float d = pos[b] - neg[b];
        const float sg = 1.0f / (1.0f + expf(-d));
        s += logf(sg);
    }
    return -s;
}

/* entity-aware regulariser (models/base.py:50-73): batch is int64 [B, 9] */
float orc_entity_reg(int64_t B, int F, const float *x, const int64_t *batch) {
    float item_s = 0.f, user_s = 0.f;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t *t = batch + b * 9;
        const float *xu = x + t[0] * F, *xi = x + t[1] * F;
        const float *ipe = x + t[3] * F, *ine = x + t[4] * F;
        const float *upe = x + t[6] * F, *une = x + t[7] * F;
        float ip = 0.f, in_ = 0.f, up = 0.f, un = 0.f;
        for (int c = 0; c < F; ++c) {
            ip += (xi[c] - ipe[c]) * (xi[c] - ipe[c]);
            in_ += (xi[c] - ine[c]) * (xi[c] - ine[c]);
            up += (xu[c] - upe[c]) * (xu[c] - upe[c]);
            un += (xu[c] - une[c]) * (xu[c] - une[c]);
        }
        const float di = (ip - in_) * (float)t[5], du = (up - un) * (float)t[8];
        item_s += logf(1.0f / (1.0f + expf(-di)));
        user_s += logf(1.0f / (1.0f + expf(-du)));
    }
    return (-item_s) + (-user_s);
}
