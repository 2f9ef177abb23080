/*
 * psh_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's k-nearest-path scan
 * (PathShadowing.shadow with Identity embedding + RelativeMSE distance +
 * PredictionContext), used ONLY as the checker in tests/, in
 * __graft_entry__.smoke() and as bench.py's cpu_baseline leg.  Nothing under
 * shadowing_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  tests/golden/make_golden.py imports the read-only
 * reference in the build container, runs shadow(cuda=False) and stores the
 * outputs; tests/test_oracle_golden.py checks this file against every one of
 * those vectors bit-for-bit (distances, indices, gathered paths).
 *
 * What is restated (reference file:line, all under shadowing/path_shadowing/):
 *   path_embedding.py:129-132,135-139   Identity embedding = conv1d with a
 *       one-hot kernel (W,1,W+h): an exact copy of y[r, t:t+W] for
 *       t in [0, T-W-h+1).  path_embedding.py:48-51 (pad_context) is what
 *       removes the last h windows.
 *   path_distance.py:62-65              RelativeMSE: ||x-y|| / ||x||.
 *       Because the embedded dataset is a permuted view (time innermost),
 *       ATen evaluates the numerator as a strided reduction:
 *           D_j = fl(x_j - y_{t+j});  acc = fma(D_j, D_j, acc), j = 0..W-1
 *           num = fl(sqrt(acc));      d = fl(num / xn)
 *       and the denominator xn = ||x|| as a contiguous reduction with 8
 *       vector lanes (see psh_oracle_qnorm).
 *   path_shadowing.py:149-173           running top-k over dataset splits:
 *       torch.topk(largest=False) twice + cat.  The reference's order among
 *       exactly equal distances is arbitrary (unstable partial sort); the
 *       oracle defines the canonical order (d asc, r asc, t asc).
 *   NON-FINITE SAMPLES (probed on the reference, tests/golden/nan_in_ensemble_*.npz): the embedding is a conv1d whose
 *       kernel -- one-hot rows for Identity, any (d,1,K) kernel otherwise -- is zero-padded by the horizon
 *       (path_embedding.py:48-51), and 0 * NaN = 0 * inf = NaN: EVERY embedded coordinate of window t is NaN as soon as ONE
 *       sample of y[t : t+K+h] -- the window or its h future samples -- is NaN or +-inf.  Such a window's distance is NaN
 *       and torch.topk(largest=False) ranks it last (path_shadowing.py:165): it never enters the top-k while k clean
 *       windows exist.  Restated below as nonfinite_prefix() / the `nf` tests.
 *   path_shadowing.py:43-58             flat index -> (r_global, t) int32.
 *   path_shadowing.py:211-216           path gather: dataset[r, t : t+W+h].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float d; int32_t r; int32_t t; } cand_t;

/* canonical strict order: (d, r, t) ascending */
static inline int cand_less(const cand_t* a, const cand_t* b) {
    if (a->d < b->d) return 1;
    if (a->d > b->d) return 0;
    if (a->r != b->r) return a->r < b->r;
    return a->t < b->t;
}

/* bounded max-heap of the k best candidates (root = current worst) */
typedef struct { cand_t* v; int n; int k; } heap_t;

static void heap_sift_down(heap_t* h, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < h->n && cand_less(&h->v[m], &h->v[l])) m = l;
        if (r < h->n && cand_less(&h->v[m], &h->v[r])) m = r;
        if (m == i) return;
        cand_t tmp = h->v[i]; h->v[i] = h->v[m]; h->v[m] = tmp;
        i = m;
    }
}
static void heap_sift_up(heap_t* h, int i) {
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!cand_less(&h->v[p], &h->v[i])) return;
        cand_t tmp = h->v[i]; h->v[i] = h->v[p]; h->v[p] = tmp;
        i = p;
    }
}
static inline void heap_offer(heap_t* h, cand_t c) {
    if (h->n < h->k) { h->v[h->n] = c; heap_sift_up(h, h->n); h->n++; return; }
    if (cand_less(&c, &h->v[0])) { h->v[0] = c; heap_sift_down(h, 0); }
}
static int cand_cmp_qsort(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (cand_less(x, y)) return -1;
    if (cand_less(y, x)) return 1;
    return 0;
}

/*
 * sum of squares of x in the order ATen's contiguous last-dim norm reduce uses
 * on x86 (path_distance.py:65, the x.norm(dim=-1) factor; probed against
 * torch 2.10 for W = 1..69): 8 lanes, lane i accumulates
 * fma(x[8b+i], x[8b+i], lane_i) over the full blocks of 8; the lanes are added
 * left to right; the W mod 8 tail then goes on top of that sum -- whole groups
 * of 4 as rounded products added one by one (no fma), the last < 4 elements as
 * a scalar fma chain.
 */
float psh_oracle_sumsq8(const float* x, int W) {
    float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int nb = W / 8;
    for (int b = 0; b < nb; ++b)
        for (int i = 0; i < 8; ++i) lane[i] = fmaf(x[8 * b + i], x[8 * b + i], lane[i]);
    float s = lane[0];
    for (int i = 1; i < 8; ++i) s = s + lane[i];
    int j = 8 * nb;
    for (; j + 4 <= W; j += 4)
        for (int i = 0; i < 4; ++i) { volatile float p = x[j + i] * x[j + i]; s = s + p; }
    for (; j < W; ++j) s = fmaf(x[j], x[j], s);
    return s;
}
float psh_oracle_qnorm(const float* x, int W) { return sqrtf(psh_oracle_sumsq8(x, W)); }

/* nf[p] = number of non-finite samples among y[0 .. p-1] (p = 0..T); returns nf[T].  Window t of a scan with field F = K + h
 * is contaminated (distance NaN) iff nf[t + F] != nf[t]. */
static int64_t nonfinite_prefix(const float* y, int64_t T, int32_t* nf) {
    int32_t c = 0;
    nf[0] = 0;
    for (int64_t p = 0; p < T; ++p) { c += !isfinite(y[p]); nf[p + 1] = c; }
    return c;
}

/* numerator accumulators for NV consecutive windows of one row (vectorisable
 * across windows; the chain over j stays sequential, as in the reference) */
#define NV 16
static inline void acc_block(const float* restrict y, const float* restrict x, int W,
                             float* restrict acc) {
    for (int v = 0; v < NV; ++v) acc[v] = 0.0f;
    for (int j = 0; j < W; ++j) {
        const float xj = x[j];
        for (int v = 0; v < NV; ++v) {
            const float D = xj - y[v + j];
            acc[v] = fmaf(D, D, acc[v]);
        }
    }
}

/* Edge case probed on the reference: with exactly ONE window per row
 * (T == W + h) the embedded dataset view collapses to a contiguous layout and
 * ATen reduces the numerator in the 8-lane order of psh_oracle_sumsq8 instead
 * of the sequential chain. */
static float acc_single_window_row(const float* y, const float* x, int W) {
    float D[4096];
    if (W > 4096) return NAN;
    for (int j = 0; j < W; ++j) D[j] = x[j] - y[j];
    return psh_oracle_sumsq8(D, W);
}

/*
 * k smallest RelativeMSE distances between each query and every admissible
 * window of every row.  dataset: R x T row-major f32.  queries: B x W.
 * qnorm: B floats or NULL (-> psh_oracle_qnorm).  out_d: B x k (ascending,
 * +inf padded), out_idx: B x k x 2 = [r_offset + r, t] (-1 padded), matching
 * path_shadowing.py:143-144.  Returns 0, or -1 on bad arguments.
 */
int psh_oracle_scan_topk(const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                         const float* queries, const float* qnorm,
                         int B, int W, int h, int k,
                         float* out_d, int32_t* out_idx, int nthreads) {
    if (!dataset || !queries || !out_d || !out_idx) return -1;
    if (R < 0 || T <= 0 || B < 0 || W <= 0 || h < 0 || k <= 0) return -1;
    const int64_t Tp = T - W - h + 1;          /* windows per row */
    if (Tp <= 0) return -1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    for (int b = 0; b < B; ++b) {
        const float* x = queries + (int64_t)b * W;
        const float xn = qnorm ? qnorm[b] : psh_oracle_qnorm(x, W);
        cand_t* all = (cand_t*)malloc(sizeof(cand_t) * (size_t)k * (size_t)nthreads);
        int* counts = (int*)calloc((size_t)nthreads, sizeof(int));
        if (!all || !counts) { free(all); free(counts); return -2; }
#pragma omp parallel num_threads(nthreads)
        {
#ifdef _OPENMP
            const int tid = omp_get_thread_num();
#else
            const int tid = 0;
#endif
            heap_t hp; hp.v = all + (size_t)tid * k; hp.n = 0; hp.k = k;
            float acc[NV];
            float tail[2 * NV + 4096];
            int32_t* nf = (int32_t*)malloc(sizeof(int32_t) * (size_t)(T + 1));
#pragma omp for schedule(dynamic, 8)
            for (int64_t r = 0; r < R; ++r) {
                const float* y = dataset + r * T;
                const int dirty = nf && nonfinite_prefix(y, T, nf) != 0;     /* rare: the row holds NaN / inf */
                for (int64_t t0 = 0; t0 < Tp; t0 += NV) {
                    int nv = (int)((Tp - t0) < NV ? (Tp - t0) : NV);
                    if (Tp == 1) {
                        acc[0] = acc_single_window_row(y, x, W);
                    } else if (nv == NV) {
                        acc_block(y + t0, x, W, acc);
                    } else if (W + NV <= (int)(sizeof(tail) / sizeof(float))) {
                        /* ragged end of the row: copy into a zero-padded buffer */
                        memset(tail, 0, sizeof(float) * (size_t)(W + NV));
                        memcpy(tail, y + t0, sizeof(float) * (size_t)(nv + W - 1));
                        acc_block(tail, x, W, acc);
                    } else {
                        for (int v = 0; v < nv; ++v) {
                            float a = 0.0f;
                            for (int j = 0; j < W; ++j) { float D = x[j] - y[t0 + v + j]; a = fmaf(D, D, a); }
                            acc[v] = a;
                        }
                    }
                    if (dirty)      /* a non-finite sample anywhere in y[t : t+W+h] makes the embedded window NaN */
                        for (int v = 0; v < nv; ++v)
                            if (nf[t0 + v + W + h] != nf[t0 + v]) acc[v] = NAN;
                    for (int v = 0; v < nv; ++v) {
                        cand_t c;
                        c.d = sqrtf(acc[v]) / xn;
                        c.r = (int32_t)(r_offset + r);
                        c.t = (int32_t)(t0 + v);
                        /* NaN never enters (all comparisons false), as with
                         * torch.topk(largest=False) which ranks NaN last */
                        if (hp.n < k) { if (c.d == c.d) heap_offer(&hp, c); }
                        else if (cand_less(&c, &hp.v[0])) heap_offer(&hp, c);
                    }
                }
            }
            counts[tid] = hp.n;
            free(nf);
        }
        /* merge the per-thread lists */
        int n = 0;
        for (int t = 0; t < nthreads; ++t) {
            if (t * k != n) memmove(all + n, all + (size_t)t * k, sizeof(cand_t) * (size_t)counts[t]);
            n += counts[t];
        }
        qsort(all, (size_t)n, sizeof(cand_t), cand_cmp_qsort);
        for (int i = 0; i < k; ++i) {
            if (i < n) {
                out_d[(int64_t)b * k + i] = all[i].d;
                out_idx[((int64_t)b * k + i) * 2 + 0] = all[i].r;
                out_idx[((int64_t)b * k + i) * 2 + 1] = all[i].t;
            } else {
                out_d[(int64_t)b * k + i] = INFINITY;
                out_idx[((int64_t)b * k + i) * 2 + 0] = -1;
                out_idx[((int64_t)b * k + i) * 2 + 1] = -1;
            }
        }
        free(all); free(counts);
    }
    return 0;
}

/*
 * The scan behind a general linear embedding (Foveal, user kernels): reference
 * path_embedding.py:117-132 (PathEmbedding.forward = conv1d with a (d,1,K) kernel;
 * Foveal's kernel :142-172) feeding RelativeMSE (path_distance.py:62-65) inside the
 * loop of path_shadowing.py:149-173.
 *   hy_i = sum_j ker[i][j] * y[t+j]   fma chain over increasing j
 *   acc  = sum_i (hx_i - hy_i)^2      D rounded, fma chain over increasing i
 *   d    = sqrt(acc) / hxnorm
 * The reference leaves both reduction orders to its libraries (MKL-DNN conv1d, the
 * vectorised norm), so THIS restatement is pinned against the reference's outputs to a
 * tolerance (tests/golden/foveal_*.npz, rtol 1e-5, indices equal outside near-ties),
 * not bit for bit; the HIP kernel is then held bit-exactly to this function.
 * ker: d x K row-major (unpadded; the context's zero taps are the integer h).
 * hx: B x d embedded queries.  hxnorm: B or NULL (-> psh_oracle_qnorm(hx, d)).
 */
/* One window per row (T == K + h): the embedded view (S, 1, d) is contiguous and the
 * numerator is reduced in the 8-lane order of psh_oracle_sumsq8 over the d coordinates,
 * as for the Identity embedding (acc_single_window_row). */
static inline float embedded_acc_one_window(const float* y, const float* ker, int d, int K, const float* hx) {
    float D[4096];
    if (d > 4096) return NAN;
    for (int i = 0; i < d; ++i) {
        const float* kr = ker + (int64_t)i * K;
        float e = 0.0f;
        for (int j = 0; j < K; ++j) e = fmaf(kr[j], y[j], e);
        D[i] = hx[i] - e;
    }
    return psh_oracle_sumsq8(D, d);
}

static inline float embedded_acc(const float* y, const float* ker, int d, int K, const float* hx) {
    float acc = 0.0f;
    for (int i = 0; i < d; ++i) {
        const float* kr = ker + (int64_t)i * K;
        float e = 0.0f;
        for (int j = 0; j < K; ++j) e = fmaf(kr[j], y[j], e);
        const float D = hx[i] - e;
        acc = fmaf(D, D, acc);
    }
    return acc;
}

int psh_oracle_scan_topk_embedded(const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                                  const float* ker, int d, int K,
                                  const float* hx, const float* hxnorm, int B, int h, int k,
                                  float* out_d, int32_t* out_idx, int nthreads) {
    if (!dataset || !ker || !hx || !out_d || !out_idx) return -1;
    if (R < 0 || T <= 0 || B < 0 || K <= 0 || d <= 0 || h < 0 || k <= 0) return -1;
    const int64_t Tp = T - K - h + 1;
    if (Tp <= 0) return -1;
    if (B == 0) return 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    /* the embedding of a window does not depend on the query: ONE pass over the ensemble, every window embedded once
     * (d fma chains over the K taps), then one chain over the d coordinates per query; a bounded heap per (thread, query) */
    cand_t* all = (cand_t*)malloc(sizeof(cand_t) * (size_t)k * (size_t)nthreads * (size_t)B);
    int* counts = (int*)calloc((size_t)nthreads * (size_t)B, sizeof(int));
    float* xn = (float*)malloc(sizeof(float) * (size_t)B);
    if (!all || !counts || !xn) { free(all); free(counts); free(xn); return -2; }
    for (int b = 0; b < B; ++b) xn[b] = hxnorm ? hxnorm[b] : psh_oracle_qnorm(hx + (int64_t)b * d, d);
    int oom = 0;
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        heap_t* hp = (heap_t*)malloc(sizeof(heap_t) * (size_t)B);
        float* e = (float*)malloc(sizeof(float) * (size_t)d);
        int32_t* nf = (int32_t*)malloc(sizeof(int32_t) * (size_t)(T + 1));
        if (!hp || !e || !nf) {
#pragma omp atomic write
            oom = 1;
        }
        if (hp) for (int b = 0; b < B; ++b) { hp[b].v = all + ((size_t)tid * B + b) * k; hp[b].n = 0; hp[b].k = k; }
#pragma omp for schedule(dynamic, 4)
        for (int64_t r = 0; r < R; ++r) {
            if (!hp || !e || !nf) continue;
            const float* y = dataset + r * T;
            const int dirty = nonfinite_prefix(y, T, nf) != 0;
            for (int64_t t = 0; t < Tp; ++t) {
                const float* yw = y + t;
                for (int i = 0; i < d; ++i) {                 /* hy_i: fma chain over increasing j */
                    const float* kr = ker + (int64_t)i * K;
                    float ev = 0.0f;
                    for (int j = 0; j < K; ++j) ev = fmaf(kr[j], yw[j], ev);
                    e[i] = ev;
                }
                const int bad = dirty && nf[t + K + h] != nf[t];     /* the zero taps of the padded kernel see it too */
                for (int b = 0; b < B; ++b) {
                    const float* x = hx + (int64_t)b * d;
                    float acc;
                    if (Tp == 1) {                             /* one window per row: the 8-lane order over d */
                        float D[4096];
                        if (d > 4096) { acc = NAN; }
                        else { for (int i = 0; i < d; ++i) D[i] = x[i] - e[i]; acc = psh_oracle_sumsq8(D, d); }
                    } else {
                        acc = 0.0f;
                        for (int i = 0; i < d; ++i) { const float D = x[i] - e[i]; acc = fmaf(D, D, acc); }
                    }
                    cand_t c;
                    c.d = bad ? NAN : sqrtf(acc) / xn[b];
                    c.r = (int32_t)(r_offset + r);
                    c.t = (int32_t)t;
                    if (hp[b].n < k) { if (c.d == c.d) heap_offer(&hp[b], c); }
                    else if (cand_less(&c, &hp[b].v[0])) heap_offer(&hp[b], c);
                }
            }
        }
        if (hp) for (int b = 0; b < B; ++b) counts[(size_t)tid * B + b] = hp[b].n;
        free(hp); free(e); free(nf);
    }
    if (oom) { free(all); free(counts); free(xn); return -2; }
    cand_t* merged = (cand_t*)malloc(sizeof(cand_t) * (size_t)k * (size_t)nthreads);
    if (!merged) { free(all); free(counts); free(xn); return -2; }
    for (int b = 0; b < B; ++b) {
        int n = 0;
        for (int t = 0; t < nthreads; ++t) {
            const int c = counts[(size_t)t * B + b];
            memcpy(merged + n, all + ((size_t)t * B + b) * k, sizeof(cand_t) * (size_t)c);
            n += c;
        }
        qsort(merged, (size_t)n, sizeof(cand_t), cand_cmp_qsort);
        for (int i = 0; i < k; ++i) {
            if (i < n) {
                out_d[(int64_t)b * k + i] = merged[i].d;
                out_idx[((int64_t)b * k + i) * 2 + 0] = merged[i].r;
                out_idx[((int64_t)b * k + i) * 2 + 1] = merged[i].t;
            } else {
                out_d[(int64_t)b * k + i] = INFINITY;
                out_idx[((int64_t)b * k + i) * 2 + 0] = -1;
                out_idx[((int64_t)b * k + i) * 2 + 1] = -1;
            }
        }
    }
    free(merged); free(all); free(counts); free(xn);
    return 0;
}

/* every window's embedded distance for one query */
int psh_oracle_all_distances_embedded(const float* dataset, int64_t R, int64_t T,
                                      const float* ker, int d, int K, const float* hx, float xn,
                                      int h, float* out /* R x Tp */) {
    const int64_t Tp = T - K - h + 1;
    if (Tp <= 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        int32_t* nf = (int32_t*)malloc(sizeof(int32_t) * (size_t)(T + 1));
        const int dirty = nf && nonfinite_prefix(dataset + r * T, T, nf) != 0;
        for (int64_t t = 0; t < Tp; ++t) {
            out[r * Tp + t] = sqrtf(Tp == 1 ? embedded_acc_one_window(dataset + r * T, ker, d, K, hx)
                                            : embedded_acc(dataset + r * T + t, ker, d, K, hx)) / xn;
            if (dirty && nf[t + K + h] != nf[t]) out[r * Tp + t] = NAN;
        }
        free(nf);
    }
    return 0;
}

/* every window's distance for one query (small cases: brute-force checks) */
int psh_oracle_all_distances(const float* dataset, int64_t R, int64_t T,
                             const float* x, float xn, int W, int h, float* out /* R x Tp */) {
    const int64_t Tp = T - W - h + 1;
    if (Tp <= 0) return -1;
    int32_t* nf = (int32_t*)malloc(sizeof(int32_t) * (size_t)(T + 1));
    for (int64_t r = 0; r < R; ++r) {
        const int dirty = nf && nonfinite_prefix(dataset + r * T, T, nf) != 0;
        for (int64_t t = 0; t < Tp; ++t) {
            float a = 0.0f;
            if (Tp == 1) a = acc_single_window_row(dataset + r * T, x, W);
            else for (int j = 0; j < W; ++j) { float D = x[j] - dataset[r * T + t + j]; a = fmaf(D, D, a); }
            out[r * Tp + t] = (dirty && nf[t + W + h] != nf[t]) ? NAN : sqrtf(a) / xn;
        }
    }
    free(nf);
    return 0;
}

/* every window's NUMERATOR acc (the value the scans compare with their admission level) for one query: the sequential
 * fp32 fma chain of path_distance.py:62-65 before the square root and the division -- tests/test_gpu_admitted_set.py
 * builds {w : acc(w) < tau} from it.  Windows the reference's zero-padded conv makes NaN are NaN here too. */
int psh_oracle_all_acc(const float* dataset, int64_t R, int64_t T, const float* x, int W, int h, float* out /* R x Tp */) {
    const int64_t Tp = T - W - h + 1;
    if (Tp <= 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        int32_t* nf = (int32_t*)malloc(sizeof(int32_t) * (size_t)(T + 1));
        const int dirty = nf && nonfinite_prefix(dataset + r * T, T, nf) != 0;
        for (int64_t t = 0; t < Tp; ++t) {
            float a = 0.0f;
            if (Tp == 1) a = acc_single_window_row(dataset + r * T, x, W);
            else for (int j = 0; j < W; ++j) { float D = x[j] - dataset[r * T + t + j]; a = fmaf(D, D, a); }
            out[r * Tp + t] = (dirty && nf[t + W + h] != nf[t]) ? NAN : a;
        }
        free(nf);
    }
    return 0;
}

int psh_oracle_all_acc_embedded(const float* dataset, int64_t R, int64_t T, const float* ker, int d, int K,
                                const float* hx, int h, float* out /* R x Tp */) {
    const int64_t Tp = T - K - h + 1;
    if (Tp <= 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        int32_t* nf = (int32_t*)malloc(sizeof(int32_t) * (size_t)(T + 1));
        const int dirty = nf && nonfinite_prefix(dataset + r * T, T, nf) != 0;
        for (int64_t t = 0; t < Tp; ++t) {
            out[r * Tp + t] = Tp == 1 ? embedded_acc_one_window(dataset + r * T, ker, d, K, hx)
                                      : embedded_acc(dataset + r * T + t, ker, d, K, hx);
            if (dirty && nf[t + K + h] != nf[t]) out[r * Tp + t] = NAN;
        }
        free(nf);
    }
    return 0;
}

/* path gather, path_shadowing.py:211-216 (single channel): out[b,i,:] =
 * dataset[r, t : t+len] for (r,t) = idx[b,i]; r is global (minus r_offset). */
int psh_oracle_gather_paths(const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                            const int32_t* idx, int64_t n, int len, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        int64_t r = idx[2 * i] - r_offset, t = idx[2 * i + 1];
        if (r < 0 || r >= R || t < 0 || t + len > T) return -1;
        memcpy(out + i * len, dataset + r * T + t, sizeof(float) * (size_t)len);
    }
    return 0;
}
