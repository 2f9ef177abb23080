// psh_kernels.h -- argument blocks and launch entry points shared by the kernels
// (psh_scan.hip, psh_embed.hip, psh_select.hip) and the C ABI (psh_capi.hip).  Internal; the public header is
// include/psh.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PSH_L 16                     // consecutive windows per lane
#define PSH_SEG (64 * PSH_L)         // windows per wave-segment
#define PSH_NSTAGE 5                 // 16-byte loads per lane per segment: ceil((SEG + W_max - 1) / 256)
#define PSH_SCAN_THREADS 1024           // one 16-wave block per CU: its waves share one work queue in LDS
#define PSH_SELECT_THREADS 1024

#define PSH_MODE_BOOT 0              // bootstrap sample: per-lane (or per-wave) minima -> buffer
#define PSH_MODE_FILTER 1            // the full scan: admit acc < tau
#define PSH_MODE_ALL 2               // exhaustive: admit every window

#define PSH_STATUS_OK_ 0
#define PSH_STATUS_OVERFLOW_ 1
#define PSH_STATUS_RETRY_ 2

namespace psh {

struct QueryState {   // one per query, device, 48 bytes
    float xn;             // ||x||
    unsigned tau_bits;    // admission threshold on acc (exclusive), float bits
    int n_valid;          // valid entries in out_d/out_idx after the last select
    float nx;             // sum of squares of the query (float, reference order)
    float thr_base;       // bound-then-verify filter: reject iff  ny - 2c > thr_base + 2^-16 * NY
    float mx_scale;       // matrix-core filter: power of two that brings data and query into f16 range (0 = unset)
    float mx_thr;         //   reject iff  t^ > mx_thr  (t^ = sum y~^2 - 2 sum x~ y~ on the scaled f16 copies)
    unsigned tau2_bits;   // ESTIMATE of the k-th smallest acc with a 2x margin (<= tau): candidates below it go to the
                          //   front of a block's slice, the rest of the admitted ones to its back (see select_kernel)
    float mx_thr2;        // the matrix-core rejection threshold that goes with tau2 (<= mx_thr)
    // the 8-bit rejection test of the batched scan (scan_mq8_kernel; zero = not armed): a segment quantised with step
    // s_y rejects a window iff  C_w - sum x^ y^  >  mx8_P / s_y + mx8_L   (integers; C_w = floor(mx8_k1 ny / s_y))
    float mx8_P, mx8_L;
    float mx8_k1;         // the same for every query of the batch (the kernel reads query 0's)
};
static_assert(sizeof(QueryState) == 48, "QueryState is 48 bytes (the host side mirrors it)");
#define PSH_UNVERIFIED_BITS 0xffffffffu   // cand_d of a back-list entry whose exact distance was not computed by the scan

#define PSH_MAX_BLOCKS 2048          // upper bound of the scan grid
#define PSH_MAX_B_PER_LAUNCH 1024    // per-block LDS append counters: one int per query

// ---- the fused single-launch scan (psh_fused.hip): state at the START of the caller's workspace -----------------
// One launch runs bootstrap -> threshold -> scan -> selection; its blocks hand data to each other through 8-byte
// {tag, value} granules and write-through stores in this header (MI355X: per-XCD L2s are not coherent, so every
// shared word is an agent-scope access).  Tags derive from `epoch`, which lives HERE (not in a kernel argument: it
// survives graph replay) and is advanced by the launch itself; psh_workspace_init writes magic + epoch once.
#define PSH_FUSED_MAGIC 0x5053484655534544ull
#define PSH_FUSED_MAX_BLOCKS 256     // blocks of the fused launch (one per CU)
#define PSH_FUSED_MAX_UNITS 4096     // bootstrap minima exchanged (one 16-byte load per thread of a block reads them all)
#define PSH_FUSED_XCD_SKEW 6          // +-2.3 %: see scan_fused_kernel
#define PSH_FUSED_FRONT 64           // candidates a block may hand to the distributed selection (~8 expected)
// the overlap-friendly three-launch step (psh_stream.hip): what its launches hand to each other.  Kernel boundaries on the
// caller's stream are the synchronisation; the two counters are device-scope atomics.
#define PSH_STREAM_MAX_Q 3           // queries one overlap-friendly step serves (their B fragments sit in LDS beside the scan's tiles)
#define PSH_STREAM_LONG_KS 18        // K-steps of 16 the shifted-query band of a long window takes at most: ceil((256 + 31) / 16)
static_assert(PSH_STREAM_LONG_KS >= PSH_STREAM_MAX_Q * 4, "bxtab holds the fragment tables of the 2-3 query step as well");
struct StreamCtl {
    unsigned ticket;                 // sample kernel: blocks that have arrived (the last one derives the levels); left at 0
    unsigned ovf;                    // scan kernel: a block met more candidates than its list holds
    unsigned armed;                  // sample kernel: 1 = the words below are valid
    unsigned scale_bits;             // the f16 scale (ONE for all queries of the step: the largest every query's proof allows)
    unsigned ncand[4];               // scan kernel: candidates appended to query q's region of FusedHdr::cand (zeroed by the sample kernel)
    unsigned tau2_bits[4], thr2_bits[4], xn_bits[4];   // per query: admission level, rejection threshold, ||x||
};
#define PSH_STREAM_CAND_CAP (PSH_FUSED_MAX_BLOCKS * PSH_FUSED_FRONT)     // 16-byte entries FusedHdr::cand holds
struct FusedHdr {
    unsigned long long magic;
    unsigned epoch;
    unsigned pad[13];
    StreamCtl stream;
    unsigned long long blk[PSH_FUSED_MAX_BLOCKS];            // end-of-scan record of a block: tag << 32 | overflow << 31 | count
    unsigned long long blk2[PSH_FUSED_MAX_BLOCKS];           //   and tag << 32 | the tau2 bits it admitted with (must agree everywhere)
    unsigned long long aflag[PSH_FUSED_MAX_BLOCKS];          // bootstrap: tag << 32 | 1 once the block's minima are written
    unsigned minima[PSH_FUSED_MAX_UNITS];                    // bootstrap: float bits of a (row, segment) minimum
    unsigned long long cand[PSH_FUSED_MAX_BLOCKS * PSH_FUSED_FRONT * 2];   // {r << 32 | d bits, t}
    // psh_stream.hip: the B fragments of the shifted query the sample kernel prepares for the scan (whose candidates go to
    // `cand` as ONE compact list of {d bits, r, t, -} entries, StreamCtl::ncand of them)
    // (one query with a LONG window, 34 <= W <= 256: [K-step][lane][8 halves] for up to PSH_STREAM_LONG_KS steps of the band)
    unsigned short bxtab[PSH_STREAM_MAX_Q * PSH_STREAM_LONG_KS * 64 * 8];   // [query][K-step][lane][8 halves]: -2 x~ shifted by the lane's column
                                                             // (a query's K-steps: 4 up to W = 33, stream_ksteps(W) for a long window)
};
#define PSH_FUSED_BYTES ((sizeof(psh::FusedHdr) + 255) / 256 * 256)

struct FusedArgs {
    FusedHdr* hdr;
    int boot_units;                  // sampled (row, segment) units: row = boot_row0 + (u / nseg) * boot_row_stride
    int64_t boot_row0, boot_row_stride;
    int rank;                        // the estimate tau2 = the rank-th smallest sampled minimum (~2k windows of the ensemble below it)
    const float* qnorm_in;           // nullable
    float* out_d;
    int32_t* out_idx;
    int* status;
    int* total;
    long long spin_ticks;            // give-up time of a poll in wall-clock ticks (100 MHz)
    int xcd_skew;                    // of the units of a pair of blocks (2j, 2j + 1) the even one takes (256 + xcd_skew) / 512
    int tbits;                       // ranking: (r, t) packs into 32 bits as r << tbits | t (-1: it does not -- the three-word compare)
    int front;                       // stream scan: candidates a block may hold (all its queries together)
    int nq;                          // stream launches: queries of the step (<= PSH_STREAM_MAX_Q)
    int units_stride;                // stream launches: query q's sampled minima start at minima[q * units_stride]
    int cand_cap;                    // stream launches: entries of a query's region of `cand_list` (query q: cand_list + q * cand_cap entries)
    void* cand_list;                 // stream launches: the queries' compact lists of admitted windows, 16-byte {d bits, r, t, q} entries
                                     // (the workspace's candidate arrays taken as one region; FusedHdr::cand on a minimal workspace)
    int k_out;                       // stream launches: row length of out_d / out_idx (= k)
    const float* tau_hint;           // nullable: the caller's admission level per query (psh_profile.tau_hint) -- no sample, no first barrier
    // a BLOCKING caller's extras (psh_shadow_blocking; the fused launch only): shadow()'s path gather (reference
    // path_shadowing.py:211-216) done by the ranking itself, and completion words the host polls instead of waiting for the
    // end of the kernel -- results and words leave as write-through system-scope stores
    const float* g_ds;               // nullable: (R, C, T) ensemble the paths of the k winners are gathered from
    float* g_out;                    //   (k, C, g_len) float32, device-ADDRESSABLE (the caller's pinned block)
    int64_t g_T;
    int g_C, g_len;
    float qv[36];                    //   the query BY VALUE (W <= 33): the launch reads it from its kernel arguments, not over PCIe
    float hint_v;                    //   with tau_hint != nullptr: the admission level by value
    unsigned* done;                  // nullable: done_shards device-addressable words; shard i's last block stores done_val there
    unsigned done_val;
    int done_shards;                 //   blocks are dealt to the shards round-robin (blockIdx % done_shards): one counter each in FusedHdr::pad
};
#define PSH_FUSED_DONE_SHARDS 8      // completion counters of a blocking call: FusedHdr::pad[4 .. 11]

struct PrepArgs {
    const float* queries;
    const float* qnorm_in;   // nullable
    int B, W;
    QueryState* qstate;
    int* total;              // B: candidates seen by the last select (diagnostics)
    int* status;             // nullable
};

struct ScanArgs {
    const float* dataset;    // R x T
    int64_t T;
    int Tp;                  // admissible windows per row
    int nseg;                // segments per row
    int W;
    int64_t row0, row_stride;
    int n_rows;              // rows visited: row0 + i * row_stride
    unsigned magic_nrs, magic_nseg;   // floor(2^32 / d) of the unit decode (fast_div)
    int64_t r_offset;
    const float* queries;    // B x W
    int B;
    int n_qgroups, q_per_group;
    int tile_floats;         // LDS floats per wave
    int k;
    QueryState* qstate;
    float* minbuf;           // B x min_stride          (BOOT)
    int64_t min_stride;
    int boot_per_wave;       // BOOT: 1 = one minimum per wave-segment, 2 = one per half segment, 0 = one per lane
    float* cand_d;           // B x cap: FILTER: one slice of `slice` entries per block;
    int2* cand_rt;           //          ALL: slot = unit * 1024 + 16 * lane + i
    int* bcount;             // B x PSH_MAX_BLOCKS: entries each block appended (FILTER)
    int* bcount2;            // nullable: entries appended to the BACK of the slice (acc in [tau2, tau)); scan_mx_kernel
    int slice;               // entries per block slice (FILTER)
    int cap;
    unsigned long long* dbg_times;   // tuning aid (nullable): per wave {start, end} wall-clock ticks (100 MHz)
    // embedded scan (ker != nullptr): windows of W = K samples are compared through a linear
    // embedding  h(y)_i = sum_j ker[i][j] * y[t + j]  against pre-embedded queries hx (B x emb_d)
    float* blockmax;         // BOOT: max |y| each block saw (feeds the f16 scale of the matrix-core filter); nullable
    int use_mx;              // FILTER: cheap test on the matrix cores (scan_mx_kernel) instead of the VALU
    int boot_wave_min;       // rows_kernel BOOT: one minimum per chunk of 64 sampled rows (minbuf[chunk]) instead of a value per row
    const void* mq_frag;     // batched matrix-core scan: B-fragment table written by the threshold kernel (f16)
    const float* ker;        // emb_d x W row-major
    const float* hx;         // B x emb_d
    int emb_d;
    int emb_wide;            // embedded scan: the 512-thread instantiation (launchers, plan)
    int emb_dense;           // embedded scan: always the dense chains, even when the kernel has suffix rows (tests, PSH_EMBED=dense)
    int emb_taps;            // embedded scan, suffix rows: walk the taps even when the support is one interval (PSH_FLAG_EMBED_TAPS)
    int emb_mx;              // embedded scan: the dense kernel's rejection test on the matrix cores (embed_mx_kernel: BOOT / FILTER)
    int boot_estimate;       // boot_mq_kernel: the minima feed an ESTIMATE of the level (values instead of upper bounds)
    int mq_i8;               // batched scan: the rejection test as an 8-bit product (scan_mq8_kernel; mq_frag holds its int8 table)
    int dbg;                 // tuning build only (PSH_DBG): timing ablations / scheduling experiments of the kernel at hand
    int emb_r1;              // prefix-sum scan: merged rows of the first phase, 0 = the plan's (tuning build: PSH_PX_R1)
    const struct EmbedPlan* plan;   // embedded scan, BOOT / FILTER: what embed_plan_kernel found in the matrix (nullable: dense chains / tap walk decided in the kernel)
};

#define PSH_EMB_MAX_D 128            // embedding rows handled natively
// What embed_plan_kernel (psh_embed_px.hip) leaves in the workspace for the two kernels launched per stage:
// contig != 0 -- every row is one constant on [a_i, ktop), one interval for all: embed_px_kernel (prefix sums) does the
// work and embed_scan_kernel returns at once; contig == 0 -- the other way round.
struct EmbedPlan {
    int contig, ktop, ngroups, d;
    float cerr_y, cerr_p;
    int r1;                          // merged rows of the scan's first phase (gtab is ordered heaviest first); == ngroups: one phase
    int pad[1];
    int4 gtab[PSH_EMB_MAX_D + 1];    // merged rows: {c' bits, byte offset of E[a_i], member rows (a byte each), members}
    // the exact verification's schedule: the rows of a survivor spread over `vnl` lanes so that the lanes' tap counts are
    // even (longest row first onto the least loaded lane); lane slot s runs vrow[vstart[s] .. vstart[s + 1])
    int vnl, vmax, vpad[2];          // lanes per survivor, taps of the busiest lane (a multiple of 4)
    int vstart[68];
    int2 vrow[PSH_EMB_MAX_D];        // {first tap & ~3 | (taps / 4) << 8 | first tap a_i << 16 | row << 24, c_i bits}
};
#define PSH_PLAN_BYTES 8192
static_assert(sizeof(EmbedPlan) <= PSH_PLAN_BYTES, "plan region");

#define PSH_EMB_MAX_TAPS 16384       // emb_d * roundup4(K) floats of LDS for the kernel matrix (what fits is decided by
                                     // the launch plan: psh_embedded_supported)
#define PSH_LDS_BYTES (160 * 1024)   // LDS per CU (gfx950)

// the batched scans' per-batch tables in the workspace (mq_frag: 512 x B4 bytes, B4 = B rounded up to 4): the SCAN's query
// table at offset 0 (written by the threshold kernel), the bootstrap's f16 copies, nx~, and the meta words
// {f16 scale, 1 / scale^2, largest ||x - s0 x^||^2 bits, largest ||x||^2 bits, largest |x| bits} (mq_prep_kernel)
#define PSH_MQ_BOOT_OFF(B4) ((size_t)192 * (size_t)(B4))
#define PSH_MQ_NX_OFF(B4) ((size_t)384 * (size_t)(B4))
#define PSH_MQ_META_OFF(B4) ((size_t)392 * (size_t)(B4))

struct ThresholdArgs {
    const float* minbuf;
    int64_t min_stride;
    int n_entries;
    QueryState* qstate;
    int k;
    int keys_in_lds;         // set by the launcher
    const float* blockmax;   // nullable: per-block max |y| of the bootstrap scan
    int n_blockmax;
    void* mq_frag;           // nullable: B-fragment table of scan_mq_kernel, (B rounded up to 4) x 256 f16
    int mq_i8;               // ... as the int8 table and per-query constants of scan_mq8_kernel instead
    int rank2;               // > 0: also estimate tau2 = the rank2-th smallest minimum (two-class candidate slices)
    PrepArgs prep;           // the per-query preparation runs here too (one launch less on the sampled path)
    const float* tau_hint;   // nullable: B levels given by the caller (psh_profile.tau_hint): no minima, no selection -- tau = tau2 = hint
};

struct SelectArgs {
    const float* cand_d;
    const int2* cand_rt;
    int64_t cand_stride;     // elements between queries
    const int* bcount;       // nullable: per-block slice counts (B x PSH_MAX_BLOCKS) -> compaction first
    const int* bcount2;      // nullable: per-block counts of the entries at the BACK of the slices (second class)
    int nblk, slice;
    int key_cap;             // distance keys that fit in LDS (set by the launcher)
    int* total;              // nullable: B, number of candidates ranked
    int n_fixed;             // flat mode: candidates per query
    int list_len;            // flat mode, gathered lists: candidate e sits at (e / list_len) * list_stride + e % list_len
    int64_t list_stride;     //   in cand_d (floats); 0 = one contiguous list per query
    int64_t list_stride_rt;  //   in cand_rt (int2 units)
    int cap;
    int k, kpad;
    int skip_negative_rows;  // merge: entries with r < 0 are padding
    int unsorted_ok;         // the k selected may be written in arbitrary order (a merge follows)
    int sort_buf_ok;         // set by the launcher: LDS holds a second kpad-item buffer behind the items (merge sort by ranking)
    uint64_t* sort_scratch;  // kpad = 16384: global second buffer of that sort, query b at + b * cand_stride (8-byte units); nullable
    int rank_sort;           // set by the launcher (kpad >= 4096, scratch at hand): the selected items go to the scratch unordered and
                             // chunk_sort_kernel / chunk_merge_kernel -- kpad / 1024 blocks per query instead of this one -- order them, write the results
    // two-class slices, fallback only: back-list entries may carry PSH_UNVERIFIED_BITS; the selection then
    // computes their exact distance itself (single query, Identity scan)
    const float* dataset;    // R x T
    const float* queries;    // B x W
    int64_t T, r_offset;
    int W;
    float* out_d;
    int32_t* out_idx;
    int2* sel_rt;            // B x kpad scratch
    int* status;             // nullable
    QueryState* qstate;      // nullable
    unsigned long long* dbg_times;   // tuning aid (nullable): phase boundaries of block 0 in wall-clock ticks (100 MHz)
    int rank_tbits;          // rank_select_kernel: (r, t) of a candidate packs into 32 bits as r << rank_tbits | t (-1: it does not)
    int* handled;            // nullable: B flags -- rank_select_kernel (one or two queries, up to PSH_RANK_CAP candidates) has written
                             // the results of query b already: select_kernel returns at once
};
#define PSH_RANK_CAP 8192            // candidates rank_select_kernel ranks (8 per thread in registers)
#define PSH_RANK_GRID 256            // its blocks per query (one or two queries; 3 .. 8 queries: 256 / B -- measured: 4 queries 176 -> 155 us per call, 8: 199 -> 189, 16: no gain): each ranks its share of the candidates against all of them
#define PSH_RANK_MAX_B 8
#define PSH_RANK_NMAX PSH_RANK_CAP    // candidates it takes in all (passes of PSH_RANK_CAP: one); a block's own share must fit PSH_RANK_OWN
#define PSH_RANK_OWN (PSH_RANK_CAP / 16 + 2)   // a block's own candidates at most

struct MergeSortedArgs {   // k best of G lists, each sorted by (d, r, t), list g holding smaller rows than list g+1
    const float* d;          // list g of query b: d + g * stride_d + b * k_in
    const int2* rt;          //                    rt + g * stride_rt + b * k_in
    int64_t stride_d, stride_rt;
    int G, k_in, k;
    float* out_d;            // B x k
    int32_t* out_idx;        // B x k x 2
};

struct ReseedArgs {      // exhaustive path: running best -> cand[b][offset .. offset + k)
    const float* out_d;
    const int32_t* out_idx;
    QueryState* qstate;
    float* cand_d;
    int2* cand_rt;
    int64_t cand_stride;
    int offset, k;
};

struct GatherArgs {
    const float* dataset;
    int64_t R, C, T, r_offset;
    const int32_t* idx;
    int64_t n;
    int64_t len;
    float* out;
};

hipError_t launch_prep(const PrepArgs& a, hipStream_t s);
hipError_t launch_qnorm(const float* q, int B, int W, float* out, hipStream_t s);
hipError_t launch_scan(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s);
hipError_t launch_embed_scan(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s);   // psh_embed.hip; launch_scan routes a.ker != nullptr here
hipError_t embed_blocks_per_cu(bool aligned, size_t shmem, int* out);
bool embed_mx_supported(int d, int K, int B, int tile_floats);          // psh_embed_mx.hip: dense kernels, rejection test on the matrix cores
hipError_t launch_embed_mx(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s);   // BOOT (half-segment minima) / FILTER
hipError_t launch_embed_plan(const float* ker, int d, int K, EmbedPlan* plan, hipStream_t s);     // psh_embed_px.hip: the structure of the matrix
bool embed_px_supported(int tile_floats, int B, int d, int K, bool wide);
hipError_t launch_embed_px(const ScanArgs& a, int mode, bool aligned, int grid, hipStream_t s);   // BOOT / FILTER; returns at once unless plan->contig
size_t scan_shmem_bytes(int tile_floats, int B, int emb_d, int W, int threads = PSH_SCAN_THREADS);
#define PSH_EMB_WIDE_MIN_B 7          // embedded scan, this many queries and more: 512-thread blocks carrying 10 (6) queries per pass
size_t scan_mx_shmem_bytes(int tile_floats, int B);
bool scan_mx_supported(int W, int B);
bool scan_mq_supported(int W, int B);          // batched queries on the matrix cores
size_t scan_mq_shmem_bytes(int tile_floats, int B);
int scan_mq_chunks(int B);                      // grid.y: chunks of queries whose fragments fit LDS
int boot_mq_chunks(int B);                      // the same for the bootstrap's fragment layout
hipError_t launch_scan_mq(const ScanArgs& a, bool aligned, int grid_x, hipStream_t s);
bool boot_mq_supported(int W);                  // bootstrap minima as matrix-core upper bounds
hipError_t launch_boot_mq(const ScanArgs& a, bool aligned, int grid_x, hipStream_t s);
hipError_t launch_mq_prep(const float* queries, int B, int W, void* mq, hipStream_t s);   // in front of launch_boot_mq: scale, padded query copies, nx~
hipError_t scan_blocks_per_cu(int W, bool aligned, bool embedded, size_t shmem, int* out);
hipError_t launch_threshold(const ThresholdArgs& a, int B, hipStream_t s);
hipError_t launch_select(const SelectArgs& a, int B, hipStream_t s);
hipError_t launch_reseed(const ReseedArgs& a, int B, hipStream_t s);
hipError_t launch_merge_sorted(const MergeSortedArgs& a, int B, hipStream_t s);   // needs G * k_in * 4 bytes of LDS
hipError_t launch_rows(ScanArgs a, int mode, int grid, hipStream_t s);     // one-window rows (T == W + h): BOOT / FILTER
size_t rows_shmem_bytes(int ds, int B);
hipError_t launch_gather(const GatherArgs& a, hipStream_t s);
bool scan_fused_supported(int W);
size_t scan_fused_shmem_bytes(int tile_floats);
hipError_t launch_scan_fused(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, hipStream_t s);
hipError_t launch_fused_init(FusedHdr* hdr, hipStream_t s);
// psh_stream.hip: the same step as three launches that can share the chip with another stream's (PSH_FLAG_OVERLAP)
size_t stream_scan_shmem_bytes(int tile_floats);
hipError_t launch_stream_sample(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, int sample_tile_floats, hipStream_t s);
hipError_t launch_stream_sample_long(const ScanArgs& a, const FusedArgs& f, int grid, hipStream_t s);   // long windows, no hint: the sample on the matrix cores
size_t stream_sample_long_shmem_bytes(int W, int nq);
hipError_t launch_stream_scan(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, hipStream_t s);
hipError_t launch_stream_rank(const ScanArgs& a, const FusedArgs& f, int grid, hipStream_t s);
size_t stream_scan_shmem_bytes_q(int tile_floats, int nq);
bool stream_long_supported(int W);              // one to three queries, 34 <= W <= 256: the scan of the step as a K-loop over the band (stream_scan_long_kernel)
size_t stream_scan_long_shmem_bytes(int W, int nq);
hipError_t launch_stream_scan_long(const ScanArgs& a, const FusedArgs& f, bool aligned, int grid, hipStream_t s);
// psh_lq.hip: batched queries with a long window (B >= 4, 34 <= W <= 256): BOOT / FILTER of the separate launches' pipeline
// the batched long-window scan's layout of a query's B fragments (psh_lq.hip), shared with the long-window sample (psh_stream.hip)
__host__ __device__ inline int lq_ksteps(int W) { return (W + 31 + 15) / 16; }
// the K-steps are compiled in as 6 / 10 / 14 / 18 (a band that ends earlier multiplies zero tables)
__host__ __device__ inline int lq_bucket(int W) { const int n = lq_ksteps(W); return n <= 6 ? 6 : (n <= 10 ? 10 : (n <= 14 ? 14 : 18)); }
__host__ __device__ constexpr int lq_rows(int nks) { return 31 + (nks + 1) / 2; }
// A query's B fragments as EIGHT SHIFTED COPIES of -2 x~ (round 6, second form) instead of one fragment per (K-step, lane): the
// fragment of lane (n = 8 a + c, hk), K-step s is -2 x~[16 s + 8 hk + i - n], i < 8 = the 16-byte chunk 2 s + hk - a + 3 of copy
// c, where copy c holds Z_c[m] = -2 x~[m - 24 - c] (zero outside the window): an ALIGNED 16-byte read at a per-lane base plus
// 32 s bytes.  A copy takes 2 NKS + 4 chunks, padded to CP = 4 mod 16 chunks so that the 16 lanes of a ds_read_b128 group
// (4 c - a takes 16 different values mod 16 for the groups' (c, a) pairs) fall on 16 different bank quads: 4.6 KB a query at
// W = 126 where the per-step fragments took 10 -- twice the queries in a chunk, half the passes over the ensemble.
__host__ __device__ constexpr int lq_copy_chunks(int nks) { return 2 * nks + 4 <= 20 ? 20 : (2 * nks + 4 <= 36 ? 36 : 52); }
__host__ __device__ constexpr int lq_query_bytes(int nks) { return 8 * lq_copy_chunks(nks) * 16; }
bool scan_lq_supported(int W, int B, int64_t T);
int scan_lq_chunk(int W, int B);                 // queries a block's chunk takes (ScanArgs::q_per_group; grid.y = ceil(B / chunk))
size_t scan_lq_shmem_bytes(int W, int B, int q_per_group);
hipError_t launch_scan_lq(const ScanArgs& a, int mode, int grid_x, hipStream_t s);
hipError_t launch_embed_rows(const float* dataset, int64_t R, int64_t T, const float* ker, int d, int K, float* out, hipStream_t s);
// psh_predict.hip: the reductions of predict_from_paths() on the device
struct MomentsArgs {
    const float* values;      // (B, k, m)
    const double* weights;    // (B, k), or nullptr: uniform 1/k
    int B, k, m;
    double* out_mean;         // (B, m)
    double* out_std;          // (B, m)
};
#define PSH_RV_MAX_T 64
struct RvArgs {
    const float* x;           // n_rows rows, row_stride floats apart
    int64_t n_rows, row_stride;
    int Ts[PSH_RV_MAX_T];     // maturities (already clipped to the row length)
    int nT, vol;
    float* out;               // (n_rows, nT)
};
// psh_prep.hip: non-finite samples the way the reference's zero-padded conv treats them
hipError_t launch_count_nonfinite(const float* x, int64_t n, unsigned long long* out, hipStream_t s);
hipError_t launch_rows_nonfinite(const float* ds, int64_t R, int64_t row_len, int* flags, hipStream_t s);
hipError_t launch_smear_nonfinite(const float* ds, int64_t R, int64_t C, int64_t T, int back, int fwd, float* out, hipStream_t s);
hipError_t launch_moments(const MomentsArgs& a, hipStream_t s);
hipError_t launch_realized_variance(const RvArgs& a, hipStream_t s);

}  // namespace psh
