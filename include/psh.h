/*
 * psh.h -- C ABI of libpsh_hip.so: the MI355X (gfx950) k-nearest-path scan.
 *
 * This is the drop-in boundary for ONE hot path of RudyMorel/shadowing:
 * PathShadowing.shadow() with the Identity embedding, the RelativeMSE distance
 * and a PredictionContext -- the sliding-window distance scan over an ensemble
 * of R trajectories followed by the top-k selection.  The reference has no FFI
 * of its own (it is pure Python on torch); the seam these entry points sit
 * under is PathShadowing.batched_distance (reference
 * shadowing/path_shadowing/path_shadowing.py:97-179) and the path gather of
 * shadow() (path_shadowing.py:211-216).  INTEGRATION.md shows the ctypes stub
 * a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every pointer marked "device" is HBM on `device`.
 *   - the caller owns every buffer (torch tensors -> data_ptr()); the library
 *     allocates nothing persistent, keeps no global state and reads no
 *     environment variable.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t, NULL = the
 *     default stream) and returns; the caller synchronises.
 *   - return value: PSH_OK or a negative PSH_ERR_*; nothing throws or aborts.
 *   - re-entrant and thread-safe given distinct workspaces.
 *   - results are bit-exact with the reference's CPU run (cuda=False): float32
 *     distances d = fl(fl(sqrt(acc))/||x||) with acc the sequential fp32 FMA
 *     chain over the window, indices (r_global, t) int32, rows sorted by
 *     (d, r, t) ascending -- the canonical order among the reference's
 *     arbitrarily ordered exact ties.
 */
#ifndef PSH_H
#define PSH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSH_VERSION 3      /* 2: psh_profile.tau_hint, psh_candidates_layout; 3: psh_shadow_blocking, psh_shadow_block_layout */

#define PSH_OK                 0
#define PSH_ERR_ARG           -1   /* NULL pointer / non-positive size / k > number of windows */
#define PSH_ERR_UNSUPPORTED   -2   /* W > PSH_MAX_W, k > PSH_MAX_K, index would overflow int32 */
#define PSH_ERR_WORKSPACE     -3   /* workspace too small (see psh_workspace_bytes) */
#define PSH_ERR_HIP           -4   /* a HIP runtime call failed (psh_last_hip_error) */
#define PSH_ERR_COMM          -5   /* RCCL could not be opened or a collective call failed (psh_last_comm_error) */

#define PSH_MAX_W      256         /* longest query window handled natively */
#define PSH_MAX_K      16384       /* largest k handled natively */

/* per-query status words written to `out_status` (device) by psh_scan_topk */
#define PSH_STATUS_OK        0
#define PSH_STATUS_OVERFLOW  1     /* candidate buffer overflowed -- or (embedded scan) fewer than k windows
                                      lay below the sampled estimate of the k-th distance: results of that
                                      query are INVALID, rerun it with the _exhaustive entry point */
#define PSH_STATUS_RETRY     2     /* the fused single-launch scan gave up (estimate short of k, a block with too many
                                      candidates, a block that was not resident, a workspace never initialised):
                                      results are INVALID, rerun the call with PSH_FLAG_NO_FUSE */

/*
 * Optional instrumentation of psh_scan_topk / psh_scan_topk_exhaustive.
 *   mode PSH_PROFILE_STAGES: HIP events are recorded on `stream` between the stages,
 *        the call SYNCHRONISES the stream and fills the *_ms fields.
 *   mode PSH_PROFILE_EVENTS: nothing is synchronised; the two caller-created
 *        hipEvent_t handles are recorded on `stream` immediately before and after the
 *        dominant kernel (the full sliding-window scan), so that a benchmark can time
 *        that kernel live inside its own timed loop.
 */
/* (PSH_PROFILE_STAGES implies the separate launches; PSH_PROFILE_EVENTS brackets the fused launch when that runs) */
#define PSH_PROFILE_STAGES 0
#define PSH_PROFILE_EVENTS 1
/* flags: the k best of every query are returned in ARBITRARY order (the sharded scan merges and
 * orders them after the all-gather anyway; saves the ordering stage of the selection kernel).
 * Honoured by psh_scan_topk on the sampled path only. */
#define PSH_FLAG_UNSORTED 1
/* A/B switches of the test-suite and the tools (results are identical either way):
 *   FILTER_VALU   the rejection test of the Identity scan on the vector ALUs instead of the matrix cores
 *   EMBED_DENSE   psh_scan_topk_embedded: dense fma chains over every tap even when the kernel has suffix rows / zero taps
 *   ROWS_GENERIC  one-window rows (T == W + h) through the generic exhaustive path instead of rows_kernel
 *   NO_FUSE       psh_scan_topk: the separate bootstrap / threshold / scan / select launches instead of the
 *                 single fused launch */
#define PSH_FLAG_FILTER_VALU  2
#define PSH_FLAG_EMBED_DENSE  4
#define PSH_FLAG_ROWS_GENERIC 8
#define PSH_FLAG_NO_FUSE      16
/* EMBED_MX: psh_scan_topk_embedded with a DENSE kernel (no suffix structure: a wavelet bank, a user kernel; d <= 12):
 * the rejection test as a split-precision banded product on the matrix cores (embed_mx_kernel), exact dense chains for the
 * survivors -- same results as without the flag.  (The library cannot look at the kernel matrix without a device
 * synchronisation, so the caller says which it is: Foveal-like kernels are faster WITHOUT the flag, on the suffix-rows path.) */
#define PSH_FLAG_EMBED_MX     64
/* EMBED_TAPS: psh_scan_topk_embedded on a suffix-rows kernel whose supports form ONE interval (Foveal): the running sums
 * by walking the taps, as for kernels with a gap, instead of differences of prefix sums (A/B tests; same results). */
#define PSH_FLAG_EMBED_TAPS   128
/* EMBED_PLAN_KEEP: psh_scan_topk_embedded: the caller's word that the previous sampled call on THIS workspace scanned with
 * the SAME kernel matrix (same contents): what that call found in the matrix (the plan region of the workspace) is used
 * again instead of being worked out by one more small launch (~25 us).  Wrong results if the matrix differs. */
#define PSH_FLAG_EMBED_PLAN_KEEP 256
/* EMBED_MX_SPLIT: with EMBED_MX: the full scan's rejection test with the three split-precision products the bootstrap uses
 * instead of one f16 product (a ten times smaller radius, three times the matrix-core work; A/B tests -- same results). */
#define PSH_FLAG_EMBED_MX_SPLIT 512
/* SELECT_ONE_BLOCK: the selection of a one- or two-query scan by the one-block radix select + sort even where the
 * ranking on all CUs applies (k <= 4096 and at most 8192 candidates; same results: A/B tests, timing). */
#define PSH_FLAG_SELECT_ONE_BLOCK 1024
/* RESERVE_CUS: the scan leaves a few compute units free: set by callers that run a collective and a merge on a side
 * stream beside the NEXT scan -- a scan otherwise owns every CU of the chip, and work on another stream would wait for
 * it (or make its last block wait).  Fused launch: grid = CUs - 4.  With PSH_FLAG_OVERLAP: grid = CUs -
 * PSH_STREAM_RESERVED_CUS, for streams made by psh_stream_create_reserving (consecutive scans overlap there, so only a
 * CU mask keeps compute units free). */
#define PSH_FLAG_RESERVE_CUS  32
/* OVERLAP: psh_scan_topk with ONE query (W <= 33): the step as three launches -- sample + admission level, scan, ranking
 * (psh_stream.hip) -- sized so that the small ones fit on the compute units BESIDE the scan of a call on ANOTHER stream, and
 * with no barrier inside the scan, so that its blocks take over a compute unit the moment the previous scan's block leaves
 * it.  For callers with independent queries in flight on two or three streams (a server; a sharded run overlapping its
 * exchange): per-step time in steady state is the ensemble's streaming time, not the fused launch's ~25 us more.  A
 * single stream is better served by the fused launch (the default).  Same results, same status protocol
 * (PSH_STATUS_RETRY -> rerun with PSH_FLAG_NO_FUSE); one workspace per stream, armed by psh_workspace_init. */
#define PSH_FLAG_OVERLAP      2048
/* MQ_F16: psh_scan_topk with a batch of queries (W <= 25): the rejection test of the batched scan as the f16 banded product
 * of rounds 2-4 (two K = 16 steps per tile) instead of the 8-bit product (one K = 32 step; since round 4 the default for
 * calls with 32 queries and more -- below that a segment's dearer set-up outweighs the halved matrix-core work).  The
 * 8-bit test puts every query of the batch on ONE quantisation step: a batch whose queries differ in amplitude by more than
 * ~4x is better served by f16 (a much smaller query keeps too many windows for its exact recheck -- slower, never wrong).
 * Same results either way (A/B tests; callers that know their batch). */
#define PSH_FLAG_MQ_F16       4096
/* LONG_LOOP: psh_scan_topk with four queries and more and a window of 34 .. 256 samples: the loop of one- to three-query steps of
 * rounds 5 (a pass over the ensemble per step) instead of the batched long-window scan (round 6: one pass per chunk of queries,
 * the queries' fragment tables in LDS, a segment's fragments in registers for all of them).  Same results (A/B tests, timing). */
#define PSH_FLAG_LONG_LOOP    8192
typedef struct psh_profile {
    int   mode;           /* in */
    int   flags;          /* in: PSH_FLAG_* */
    void* ev_scan_begin;  /* in (PSH_PROFILE_EVENTS): hipEvent_t */
    void* ev_scan_end;    /* in (PSH_PROFILE_EVENTS): hipEvent_t */
    float prep_ms;        /* query norms + state reset                      */
    float sample_ms;      /* sample-rows scan -> histogram of lane minima   */
    float threshold_ms;   /* histogram -> admission threshold               */
    float scan_ms;        /* the full sliding-window scan + filter (HBM-bound kernel) */
    float select_ms;      /* radix select + bitonic sort of the survivors   */
    float total_ms;
    int   path;           /* 0 = sampled threshold path, 1 = exhaustive path, 2 = sampled path as ONE fused launch,
                             3 = the three overlap-friendly launches (PSH_FLAG_OVERLAP) */
    int   n_sample_rows;
    int   grid_blocks;    /* blocks of the scan kernel */
    int   n_candidates;   /* PSH_PROFILE_STAGES: largest per-query candidate count the scan admitted */
    /* in (version 2), optional: the caller's ADMISSION HINT -- device-addressable, B floats, or NULL.
     * hint[b] is a level on acc = sum_j (x_j - y_{t+j})^2 = (d ||x||)^2 (behind an embedding: sum_i (hx_i - hy_i)^2): the call
     * admits the windows of query b with acc < hint[b] and takes NO bootstrap sample (no sample launch; the fused launch skips
     * its sample phase and its first grid barrier).  The results are the exact top-k whenever at least k windows lie below
     * the hint; when fewer do -- or the hint is not a positive finite number, or it admits more than the candidate lists hold --
     * the query's status says PSH_STATUS_OVERFLOW / PSH_STATUS_RETRY as for a sampled level that fell short, and the caller
     * reruns the call WITHOUT the hint.  Meant for consecutive queries whose k-th distance is known roughly (rolling query
     * dates: the previous call's out_d[b][k-1] -> hint = (d_k ||x||)^2 x margin), and for tests that pin the admitted set
     * (psh_candidates_layout).  The small-problem / exhaustive paths ignore it.  Mind the window length: the number of windows
     * below a level grows like level^(W / 2), so a margin of 10 % on acc admits ~2.6 k windows for k = 1024 at W = 20 and
     * hundreds of thousands at W = 126 (-> slow, or PSH_STATUS_RETRY beyond the lists' 65536 entries): margin ~ 2.6^(2 / W). */
    const float* tau_hint;
} psh_profile;

int         psh_version(void);
const char* psh_strerror(int code);
const char* psh_last_hip_error(void);   /* text of the last failing HIP call on this thread */

/*
 * Bytes of device workspace psh_scan_topk / psh_scan_topk_exhaustive want for
 * this problem (a larger workspace is used as a larger candidate buffer).
 */
int psh_workspace_bytes(int64_t R, int64_t T, int B, int W, int h, int k, size_t* out_bytes);
/*
 * Arm the fused single-launch scan for this workspace: psh_scan_topk with ONE query (W <= 33) then runs bootstrap,
 * threshold, scan and selection in one launch whose blocks exchange data through a header at the start of the
 * workspace.  Call once after allocating the workspace (and again after a PSH_STATUS_RETRY caused by a time-out);
 * a workspace that was never initialised is detected on the device and the call reports PSH_STATUS_RETRY.
 * The header keeps an epoch across launches: one workspace serves ONE stream at a time.
 */
int psh_workspace_init(int device, void* stream, void* workspace, size_t workspace_bytes);

/*
 * ||x||_2 of each query in the reduction order of the reference's
 * x.norm(dim=-1) (path_distance.py:65).  queries: device B x W.  out: device B.
 */
int psh_query_norm(int device, void* stream, const float* queries, int B, int W, float* out_qnorm);

/*
 * The scan: k nearest windows of every query.  Replaces the loop body of
 * PathShadowing.batched_distance (path_shadowing.py:149-173: embed, distance,
 * topk, index decode, running merge) for Identity + RelativeMSE +
 * PredictionContext(horizon = h).
 *
 *   dataset   device, R x T row-major float32 (single channel), resident in HBM
 *   r_offset  added to the row index in out_idx (shard -> global row)
 *   queries   device, B x W float32
 *   qnorm     device, B floats, or NULL: computed as psh_query_norm does
 *   h         PredictionContext.horizon (0 for None): windows t in [0, T-W-h]
 *   out_d     device, B x k float32, ascending
 *   out_idx   device, B x k x 2 int32 = [r_offset + r, t]
 *   out_status device, B int32 (PSH_STATUS_*)
 *
 * "device" for the small arguments (queries, out_d, out_idx, out_status; paths of psh_gather_paths) means device-ADDRESSABLE:
 * pinned host memory mapped into the device's address space (hipHostMalloc) is fine -- a blocking caller can have the kernels
 * read its query from and write its results into host memory and spare two copies (what shadowing_amd's shadow() does for one
 * query: 80 bytes in, B*k*12 bytes + the gathered paths out).  The ensemble and the workspace belong in HBM.
 * Requires R*(T-W-h+1) >= k (the reference raises for k larger than a split,
 * path_shadowing.py:165) and r_offset + R, T < 2^31.
 * One-window rows (T == W + h, e.g. N pre-embedded points of PathDistance.forward_topk with
 * W = T = d): the reference's numerator is then the contiguous 8-lane reduce, and the scan
 * runs a row per lane (rows_kernel) instead of a segment per wave.
 *
 * STATUS PROTOCOL -- every caller looks at out_status before it uses the results (PSH_FLAG_NO_FUSE or not, whatever B):
 *   PSH_STATUS_OK        out_d / out_idx of that query are the exact answer.
 *   PSH_STATUS_OVERFLOW  that query's candidate slices overflowed (massive exact ties) or fewer than k windows lay below the
 *                        sampled estimate: its results are INVALID -> psh_scan_topk_exhaustive for that query.
 *   PSH_STATUS_RETRY     the launches that serve 1, 2 or 3 queries (W <= 33) -- the fused single launch for one query,
 *                        the three overlap-friendly launches for one query under PSH_FLAG_OVERLAP and for EVERY call with
 *                        B = 2 or 3, flag or no flag -- admit below a statistical estimate and give up when it falls short
 *                        of k, when the candidate lists overflow (the fused launch: 64 entries per block; the three
 *                        launches: a query's list in the workspace, out[13] of psh_candidates_layout), or on a workspace
 *                        psh_workspace_init never armed:
 *                        results of EVERY query of the call are INVALID -- since version 2 the launches overwrite them with NaN
 *                        distances and (-1, -1) indices instead of leaving an earlier call's numbers -> the same call with PSH_FLAG_NO_FUSE
 *                        (the separate launches: a provable bound, per-query OVERFLOW as above).
 *   A call with psh_profile.tau_hint whose status is not OK for some query: the hint fell short (or was useless) -> the
 *                        same call WITHOUT the hint, then as above.
 * shadowing_amd/_native.py: scan_topk_checked is this protocol in 20 lines.
 *
 * WINDOW LENGTHS.  The rejection test of a scan runs on the matrix cores for ONE query with W <= 256 (W <= 33: the fused
 * launch / the three overlap-friendly launches with the shifted-query band in registers; 34 <= W <= 256: the three launches
 * with the band as a K-loop over ceil((W + 46) / 16) steps, flag or no flag -- stream_scan_long_kernel; round 6: the window
 * energies from fp32 prefix sums, one MFMA a step), for two or three queries with W <= 33 or a long window (they ride one pass
 * of the three launches: three up to W = 97, two up to W = 145 -- what fits LDS), for
 * larger batches with W <= 25, for four queries and more with 26 <= W <= 256 and for two or three beyond what rides one pass --
 * round 6: the batched long-window scan (psh_lq.hip) through the separate launches, one pass per chunk of queries;
 * PSH_FLAG_LONG_LOOP: round 5's loop of two- or three-query steps inside the call -- (status words per query as always);
 * PSH_FLAG_FILTER_VALU / PSH_FLAG_NO_FUSE calls use the vector-ALU filter or the exact chains.  Results do not depend on
 * which.
 */
int psh_scan_topk(int device, void* stream,
                  const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                  const float* queries, const float* qnorm, int B, int W, int h, int k,
                  float* out_d, int32_t* out_idx, int32_t* out_status,
                  void* workspace, size_t workspace_bytes, psh_profile* profile);

/*
 * Same contract, no sampling and no admission threshold: the dataset is
 * processed in row chunks whose every window fits the candidate buffer.
 * Always exact (any amount of ties), several times slower.  out_status is
 * always PSH_STATUS_OK.
 */
int psh_scan_topk_exhaustive(int device, void* stream,
                             const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                             const float* queries, const float* qnorm, int B, int W, int h, int k,
                             float* out_d, int32_t* out_idx, int32_t* out_status,
                             void* workspace, size_t workspace_bytes, psh_profile* profile);

/*
 * The scan behind a general LINEAR embedding (Foveal, any PathEmbedding kernel):
 * replaces the same loop body (path_shadowing.py:149-173) when the embedding is
 * conv1d with a (d, 1, K) kernel (path_embedding.py:117-132, Foveal :142-172) and the
 * distance is RelativeMSE.  For every window t in [0, T-K-h] of every row
 *     hy_i = sum_j kernel[i][j] * y[t+j]         i < d   (fma chain, increasing j)
 *     acc  = sum_i (hx_i - hy_i)^2                        (fma chain, increasing i)
 *     d    = sqrt(acc) / hxnorm
 *   kernel  device, d x K row-major float32 (the UNPADDED kernel: the context's zero
 *           taps over the horizon are the integer h)
 *   hx      device, B x d float32: the embedded queries (embedding(x_context))
 *   hxnorm  device, B floats, or NULL: ||hx|| in the order of psh_query_norm
 * Everything else as psh_scan_topk (same workspace: psh_workspace_bytes with W = K).
 * The reference evaluates these sums in library-chosen orders, so agreement with IT is
 * to ~1e-6 relative (tested at 1e-5) with indices equal outside near-ties; agreement
 * with the oracle's restatement of the order above is bit-exact.  Requires d <= 128, K <= PSH_MAX_W and a
 * kernel matrix that fits LDS beside the wave tiles (psh_embedded_supported: up to 64 x 256, or 39 x 252 --
 * Foveal(1.15, 0.9, 252)), finite data, and MORE than one window per row:
 * T == K + h returns PSH_ERR_UNSUPPORTED -- that layout is scanned as R pre-embedded points, psh_embed_rows
 * followed by psh_scan_topk (below).
 * Kernels whose rows are one constant each on a common support from some tap onwards
 * (Foveal, also with an ImputationContext's gap; recognised on the device, K <= 256) are
 * scanned bound-then-verify over shared running sums -- same results, ~5x faster; when the
 * common support is one interval (Foveal itself) the running sums are differences of prefix sums
 * of the segment -- another ~2x (PSH_FLAG_EMBED_TAPS: the tap walk instead).
 * PSH_FLAG_EMBED_DENSE forces the dense chains over ALL K taps of every row, zero taps included (A/B tests; and the
 * exhaustive pass over rows that hold non-finite samples: 0 * NaN = NaN as in the reference's conv -- psh_rows_nonfinite).
 */
int psh_embedded_supported(int d, int K);   /* 1 when a d x K kernel fits the embedded scan (LDS), else 0 */
/* Diagnostics: byte offset, inside the workspace, of what the last sampled psh_scan_topk_embedded call found in its kernel
 * matrix -- four int32 {one_interval, ktop, merged_rows, d}: one_interval != 0 says the prefix-sum scan did the work
 * (tests and tools read it after synchronising the stream). */
size_t psh_embed_plan_offset(void);
/* Diagnostics (tests, tools): where a psh_scan_topk / psh_scan_topk_embedded call with these sizes on a workspace of
 * `workspace_bytes` leaves the windows its scan ADMITTED (acc below the level), before the selection picks k of them:
 *   out[0] byte offset of the per-query state (48 bytes a query: ||x||, the level's float bits, ...)
 *   out[1] byte offset of bcount  (int32 [B][out[9]]: entries block i appended to the FRONT of its slice of query b)
 *   out[2] byte offset of bcount2 (int32 [out[9]]: single-query matrix-core scan with two-class slices: entries at the BACK)
 *   out[3] byte offset of cand_d  (float [B][cap]: distances; 0xffffffff = a back entry whose distance was not computed)
 *   out[4] byte offset of cand_rt (int32 [B][cap][2]: (r_global, t))
 *   out[5] cap (entries per query); block i's slice of query b starts at b * cap + i * (cap / grid_blocks) -- grid_blocks
 *          from psh_profile after the call -- front entries upwards from its start, back entries downwards from its end
 *   out[6] byte offset of the fused / overlap-friendly launches' candidate area (16-byte entries), out[7] of the fused
 *          launch's per-block records (uint64 [out[10]]: low 31 bits = entries of block i, which start at entry i * out[11];
 *          an entry = {d bits | r << 32, t}), out[8] of the overlap-friendly launches' per-query counts (uint32 [4])
 *   out[9] PSH_MAX_BLOCKS, out[10] blocks of the fused launch at most, out[11] entries a fused block may publish
 *   out[12] byte offset of the overlap-friendly launches' lists (16-byte entries {d bits, r, t, q}; query q's list starts at
 *          entry q * out[13]), out[13] entries a query's list holds: the region of cand_d / cand_rt taken as one array, at
 *          most 65536 per query (a block whose own 64-entry list is full appends there directly: clustered matches in a
 *          smooth ensemble), or out[6] with out[10] * out[11] / B entries on a workspace too small for more
 * Nothing is launched; PSH_ERR_WORKSPACE when the workspace is too small for the sizes. */
int psh_candidates_layout(int64_t R, int64_t T, int B, int W, int h, int k, size_t workspace_bytes, int64_t* out14);
int psh_scan_topk_embedded(int device, void* stream,
                           const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                           const float* kernel, int d, int K,
                           const float* hx, const float* hxnorm, int B, int h, int k,
                           float* out_d, int32_t* out_idx, int32_t* out_status,
                           void* workspace, size_t workspace_bytes, psh_profile* profile);
int psh_scan_topk_embedded_exhaustive(int device, void* stream,
                           const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                           const float* kernel, int d, int K,
                           const float* hx, const float* hxnorm, int B, int h, int k,
                           float* out_d, int32_t* out_idx, int32_t* out_status,
                           void* workspace, size_t workspace_bytes, psh_profile* profile);

/*
 * One-window rows behind a linear embedding (T == K + h).  The reference's embedded view (S, 1, d) is then
 * contiguous and RelativeMSE reduces over d in the 8-lane order (path_embedding.py:129-132, path_distance.py:65),
 * exactly what psh_scan_topk computes for rows ONE window long.  So:
 *     psh_embed_rows(dataset R x T, kernel d x K)  ->  out R x d:  out[r][i] = sum_j kernel[i][j] * y[r][j]
 *                                                                  (fma chain over increasing j)
 *     psh_scan_topk(dataset = out, R, T = d, queries = hx (B x d), W = d, h = 0)      (t is 0 in every index)
 * out: device, R x d float32, caller-owned.
 */
int psh_embed_rows(int device, void* stream, const float* dataset, int64_t R, int64_t T,
                   const float* kernel, int d, int K, float* out);

/*
 * Merge G sorted-or-not candidate lists per query into the k best by (d, r, t):
 * the running merge of path_shadowing.py:170-173 and the cross-GPU merge after
 * the all-gather of per-shard results.
 *   d_lists   device, B x n_in float32   (n_in = G * k_in, lists concatenated)
 *   idx_lists device, B x n_in x 2 int32 (entries with r < 0 are padding)
 *   workspace device, >= psh_merge_workspace_bytes(B, k)
 */
int psh_merge_workspace_bytes(int B, int k, size_t* out_bytes);
/* The same for G lists that sit where an all-gather left them, without a repacking copy:
 * list g of query b starts at d_gathered + g*rank_stride + b*k_in (floats) and at
 * idx_gathered + 2*(g*rank_stride_idx + b*k_in) (int32; the stride counts (r,t) pairs). */
int psh_merge_topk_gathered(int device, void* stream,
                            const float* d_gathered, const int32_t* idx_gathered,
                            int G, int64_t rank_stride, int64_t rank_stride_idx, int B, int k_in, int k,
                            float* out_d, int32_t* out_idx, void* workspace, size_t workspace_bytes);
/* The same when every list arrives SORTED by (d, r, t) (what psh_scan_topk returns) and list g holds
 * smaller rows than list g+1 (ranks own ascending row blocks): no selection, no sort -- each entry's
 * merged position is its own position plus a binary-search count per other list.  No workspace.
 * Requires G <= 64 and G * k_in <= 32768 (the distance keys sit in LDS); PSH_ERR_UNSUPPORTED otherwise. */
int psh_merge_sorted_gathered(int device, void* stream,
                              const float* d_gathered, const int32_t* idx_gathered,
                              int G, int64_t rank_stride, int64_t rank_stride_idx, int B, int k_in, int k,
                              float* out_d, int32_t* out_idx);
int psh_merge_topk(int device, void* stream,
                   const float* d_lists, const int32_t* idx_lists, int B, int n_in, int k,
                   float* out_d, int32_t* out_idx, void* workspace, size_t workspace_bytes);

/*
 * ---- multi-GPU: the exchange of the row-sharded scan (the reference has no multi-GPU path) ------------
 * One process per GPU.  Rank g keeps rows [lo_g, hi_g) resident, scans them with r_offset = lo_g (psh_scan_topk),
 * and the per-rank top-k lists meet in ONE all-gather of B*k*12 bytes per rank (RCCL over xGMI), after which every
 * rank merges the G sorted lists with the same (d, r, t) order -- results identical for any G.
 * RCCL is opened at run time from `librccl_path` (NULL: "librccl.so"); hand over the library the process already
 * uses (PyTorch-ROCm's torch/lib/librccl.so) so that there is one RCCL in the process.
 *   psh_comm_unique_id   rank 0 creates the 128-byte id; the caller broadcasts it by its own means
 *   psh_comm_create      collective over the `world` ranks (ncclCommInitRank)
 *   psh_exchange_merge   enqueues, without synchronising the host:
 *                          compute stream: record ev_scan_done            (the local scan wrote `send` before it)
 *                          side stream   : wait ev_scan_done; all-gather send -> gathered; merge -> out_d / out_idx;
 *                                          record ev_merged
 *                        so the collective and the merge run beside whatever the compute stream does next (the scan
 *                        of the next query batch); the consumer of out_d / out_idx waits for ev_merged.
 *     send      device, 3*B*k int32: the (B,k) float32 distances (bit patterns), then the (B,k,2) int32 indices --
 *               psh_scan_topk writes straight into it (out_d = send, out_idx = send + B*k)
 *     gathered  device, G x 3*B*k int32 (the all-gather's receive buffer, caller-owned)
 *     merge_workspace  psh_merge_workspace_bytes(B, k); only used when the sorted merge does not apply (G > 64 or
 *               G*k > 32768)
 *     ev_scan_done, ev_merged   hipEvent_t created by the caller
 *   Requires B*k even.  Buffers must stay alive until ev_merged has completed.
 */
#define PSH_COMM_ID_BYTES 128
typedef struct psh_comm psh_comm;
const char* psh_last_comm_error(void);
int psh_comm_unique_id(const char* librccl_path, void* out_id);
int psh_comm_create(const char* librccl_path, int device, int world, int rank, const void* id, psh_comm** out);
int psh_comm_destroy(psh_comm* comm);
int psh_comm_world(const psh_comm* comm);
int psh_exchange_merge(psh_comm* comm, void* compute_stream, void* side_stream,
                       const int32_t* send, int32_t* gathered, int B, int k,
                       float* out_d, int32_t* out_idx, void* merge_workspace, size_t merge_workspace_bytes,
                       void* ev_scan_done, void* ev_merged);

/*
 * A HIP stream whose kernels never run on `reserve_cus` of the device's compute units (hipExtStreamCreateWithCUMask): for
 * callers that keep a collective and a merge in flight on ANOTHER stream beside their scans.  A scan block holds its
 * compute unit for ~75 us; a foreign workgroup that does not fit into what the scan leaves free (one wave slot, 64 VGPRs
 * per SIMD, 16 KB of LDS) would otherwise wait for a block to leave -- and delay the next scan's block there by as much.
 * Scans on such a stream must be issued with PSH_FLAG_RESERVE_CUS (their grid then matches the compute units they can
 * use).  out_reserved receives the number actually reserved (0 when the device has too few).  The stream belongs to the
 * caller: psh_stream_destroy (or hipStreamDestroy).  (The reference has no counterpart: its multi-GPU story is absent.)
 */
#define PSH_STREAM_RESERVED_CUS 8
int psh_stream_create_reserving(int device, int reserve_cus, void** out_stream, int* out_reserved);
int psh_stream_destroy(int device, void* stream);

/*
 * Path gather of shadow() (path_shadowing.py:211-216):
 *   out[i, c, :] = dataset[idx[i,0] - r_offset, c, idx[i,1] : idx[i,1] + len]
 * dataset: device R x C x T;  idx: device n x 2;  out: device n x C x len.
 * Entries whose row is outside [0, R) (other shards' rows, padding) are left
 * untouched.
 */
int psh_gather_paths(int device, void* stream,
                     const float* dataset, int64_t R, int64_t C, int64_t T, int64_t r_offset,
                     const int32_t* idx, int64_t n, int len, float* out);

/*
 * ONE BLOCKING shadow() OF ONE Identity QUERY -- the reference's own call (README.md:47-57; path_shadowing.py:181-218:
 * normalise, batched_distance, the path gather, .cpu().numpy()) as ONE library call that returns when the results ARE in the
 * caller's host memory.  The only entry point that waits: it is the latency of this call that a caller of the reference's
 * API sees, and it is dominated by what lies around the ~100 us scan -- a second launch for the gather, the copies out, the
 * runtime's end-of-kernel notification.  Here
 *   - the query (and the optional admission hint) are read by the kernels from `host_block`, and the k distances, indices
 *     and gathered paths are WRITTEN BY THE KERNELS straight into it (write-through stores over PCIe: 172 KB at k = 1024);
 *   - when the fused single launch serves the call (W <= 33, the sizes psh_scan_topk takes it for), the ranking phase of that
 *     launch gathers the winners' paths itself -- no psh_gather_paths launch -- and its last blocks set completion words in the
 *     block which this function polls: it returns ~1 us after the last result byte has landed instead of after the
 *     runtime has noticed the kernel's end;
 *   - otherwise (long windows, large k, small ensembles): psh_scan_topk's launches + the gather launch, then
 *     hipStreamSynchronize.
 *   host_block        pinned host memory the device can address (hipHostMalloc / torch pin_memory; fine-grained, the HIP
 *                     default), >= out7[0] bytes of psh_shadow_block_layout; the CALLER writes the W query samples at byte
 *                     PSH_SHADOW_OFF_QUERY (and, with_hint != 0, the admission level -- psh_profile.tau_hint's meaning -- at
 *                     PSH_SHADOW_OFF_HINT) before the call and reads after it: int32 status at out7[1] (PSH_STATUS_*, the
 *                     STATUS PROTOCOL of psh_scan_topk: anything but OK -> the results are invalid, rerun through
 *                     psh_scan_topk with PSH_FLAG_NO_FUSE / without the hint), float32 d[k] at out7[4], int32 idx[k][2] at
 *                     out7[5], float32 paths[k][C][W + h] at out7[6].  The words in between belong to the library.
 *   rows              device, R x T: what the scan reads (channel 0; the smeared rows of an ensemble with non-finite samples)
 *   dataset3          device, R x C x T: what the paths are gathered from (== rows when C == 1)
 *   workspace         as psh_scan_topk's (B = 1), armed by psh_workspace_init; one workspace and one host block per stream
 *   profile           optional: flags in, path / grid_blocks out (path 2 = the fused launch served the call); its tau_hint and
 *                     events are ignored (with_hint selects the hint in the block)
 * Returns PSH_OK when the call has COMPLETED on the device (whatever the status word says), else a PSH_ERR_*.
 */
#define PSH_SHADOW_OFF_STATUS 0
#define PSH_SHADOW_OFF_DONE   64     /* 8 completion words + the call counter at 96: the library's */
#define PSH_SHADOW_OFF_SEQ    96
#define PSH_SHADOW_OFF_TIMES  16     /* diagnostics, 4 floats: us the last call spent enqueueing / waiting / until its launch had started / until the first completion word (fused launch) */
#define PSH_SHADOW_OFF_STARTED 60     /* diagnostics: the launch's first block sets it to the call counter when it starts */
#define PSH_SHADOW_OFF_HINT   112
#define PSH_SHADOW_OFF_QUERY  128    /* PSH_MAX_W floats */
int psh_shadow_block_layout(int W, int h, int k, int64_t C, size_t* out7);   /* {bytes, status, query, hint, d, idx, paths} */
int psh_shadow_blocking(int device, void* stream,
                        const float* rows, int64_t R, int64_t T, int64_t r_offset,
                        const float* dataset3, int64_t C, int W, int h, int k,
                        void* host_block, size_t host_block_bytes, int with_hint,
                        void* workspace, size_t workspace_bytes, psh_profile* profile);

/*
 * NON-FINITE SAMPLES.  The scans above judge a window by its own samples -- psh_scan_topk by all W of them,
 * psh_scan_topk_embedded by the taps [lo, hi) that the span of some kernel row covers (first to last tap with a non-zero
 * entry in any row) -- a NaN among those: distance NaN, never returned while k clean windows exist.  The reference's
 * embedding is a conv1d whose kernel is ZERO-PADDED by the horizon (path_embedding.py:48-51), and 0 * NaN = 0 * inf =
 * NaN: there a window is NaN as soon as one sample of y[r, :, t : t+K+h] -- the window, its h future samples, any channel
 * -- is NaN or +-inf.  A caller that wants exactly that for an ensemble holding such samples scans rows in which every
 * non-finite sample has been written over the h + (K - hi) samples before it and the lo samples after it:
 *   psh_count_nonfinite  *out_count (device, 8 bytes) = number of NaN / +-inf among n floats: 0 (almost always) -> scan
 *                        the ensemble as it is;
 *   psh_smear_nonfinite  out[r, q] (device R x T) = NaN if any channel of dataset (device R x C x T) holds a non-finite sample in
 *                        [q - fwd, q + back], else dataset[r, 0, q]; pass `out` as the scan's dataset with back = h + K - hi,
 *                        fwd = lo, gather paths from the original.  shadowing_amd.PathShadowing does this once per
 *                        resident ensemble.
 */
int psh_count_nonfinite(int device, void* stream, const float* x, int64_t n, unsigned long long* out_count);
/*   psh_rows_nonfinite   out_flags[r] (device, R int32) = 1 if row r of dataset (device R x C x T) holds a non-finite sample in any
 *                        channel, else 0.  For the EMBEDDED scans, whose rejection tests assume finite data (prefix sums and
 *                        matrix-core tiles spread a NaN over clean windows): a caller scans the clean rows with
 *                        psh_scan_topk_embedded, the dirty ones -- smeared with back = h, fwd = 0 -- with
 *                        psh_scan_topk_embedded_exhaustive (dense chains over all K taps: 0 * NaN = NaN exactly where the
 *                        reference's zero-padded conv has it), and merges the two lists (psh_merge_topk).
 *                        shadowing_amd.PathShadowing does this. */
int psh_rows_nonfinite(int device, void* stream, const float* dataset, int64_t R, int64_t C, int64_t T, int32_t* out_flags);
int psh_smear_nonfinite(int device, void* stream, const float* dataset, int64_t R, int64_t C, int64_t T, int back, int fwd, float* out);

/*
 * The reductions of predict_from_paths() (path_shadowing.py:245-252: `proba.avg(values, axis=1)`,
 * `proba.std(values, axis=1)` over the k shadowing paths):
 *   out_mean[b, i] = sum_j w[b, j] * values[b, j, i]
 *   out_std [b, i] = sqrt(sum_j w[b, j] * (values[b, j, i] - out_mean[b, i])^2)
 * values: device B x k x m float32 (the statistic `to_predict` evaluated on the out-context of the k paths);
 * weights: device B x k float64 AS GIVEN (the averaging class's, normalised or not -- nothing is renormalised), or NULL:
 * uniform 1/k;  out_mean, out_std: device B x m float64.  Sums in double, fixed order.
 */
int psh_weighted_moments(int device, void* stream, const float* values, const double* weights, int B, int k, int m,
                         double* out_mean, double* out_std);

/*
 * realized_variance (shadowing/statistics.py:5-16) of n_rows rows of log-returns:
 *   out[r, i] = mean(x[r, :Ts[i]]^2) * 252        (its square root if vol != 0)
 * x: device, row r starts at x + r * row_stride and holds `len` samples (a view into gathered paths: row_stride = W + h,
 * x = paths + W, len = h);  Ts: HOST array of nT <= 64 maturities in samples (clipped to len, as numpy's slice is);
 * out: device n_rows x nT float32.  Squares in fp32, sums in double.
 */
int psh_realized_variance(int device, void* stream, const float* x, int64_t n_rows, int64_t row_stride, int len,
                          const int* Ts, int nT, int vol, float* out);

#ifdef __cplusplus
}
#endif
#endif /* PSH_H */
