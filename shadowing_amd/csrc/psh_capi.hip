#include <cmath>
// psh_capi.hip -- the C ABI declared in include/psh.h: argument checking, workspace
// carving, launch planning.  Everything is enqueued on the caller's stream; no
// allocation, no global state (a thread-local string holds the last HIP error text).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "psh.h"
#include "psh_kernels.h"

using namespace psh;

namespace {

thread_local char g_hip_err[256] = "";

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            snprintf(g_hip_err, sizeof(g_hip_err), "%s -> %s (%s:%d)", #expr,                \
                     hipGetErrorString(e__), __FILE__, __LINE__);                            \
            return PSH_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

struct DeviceGuard {   // launch on the caller's device, restore the previous one
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != device && hipSetDevice(device) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int next_pow2(int x) { int p = 2; while (p < x) p <<= 1; return p; }
inline int tile_floats_for(int W) {
    const int logical = PSH_SEG + W + 32;               // SEG + W - 1 used, slack for the sliding refills
    return (logical + ((logical >> 6) << 2) + 8 + 3) & ~3;
}

struct Workspace {
    FusedHdr* fused;      // state of the fused single-launch scan: ALWAYS the first PSH_FUSED_BYTES of the workspace
    EmbedPlan* eplan;     // embedded scan: what embed_plan_kernel found in the kernel matrix (PSH_PLAN_BYTES)
    QueryState* qstate;
    int* total;
    float* minbuf;
    int64_t min_stride;
    int* bcount;
    int* bcount2;         // PSH_MAX_BLOCKS ints: second-class counts of the single-query matrix-core scan
    float* blockmax;      // PSH_MAX_BLOCKS floats: per-block max |y| of the bootstrap scan
    void* mq_frag;        // (B rounded up to 4) x 256 f16: B fragments of the batched matrix-core scan
    int2* sel_rt;
    float* cand_d;
    int2* cand_rt;
    int cap;
    int kpad;
};

// bootstrap sample: which rows (spread over the ensemble) feed the admission threshold,
// and at which granularity their minima are kept.  The scan then admits about
// k * R / rows windows.
//   per-wave mode: one minimum per 1024-window segment, 1/16 of the rows (more, up to a
//                  quarter, until there are >= 8k of them);
//   half-segment mode (per_wave = 2; the single-query matrix-core scan, whose two-class slices keep the
//                  selection cheap whatever tau admits): two minima per segment, half the rows -- the
//                  bootstrap is a latency-bound launch and one segment per wave is one round trip less;
//   per-lane mode: one minimum per 16 windows, for ensembles too small for the above;
//   rows == 0    : sample too thin to be useful -> exhaustive path.
struct BootPlan { int64_t rows; int per_wave; int64_t entries; };
// `estimate`: the scan admits below the sampled ESTIMATE of the k-th distance (embedded scan), so the sample only has
// to carry that estimate's rank (~2k * sampled fraction) comfortably -- 2k minima instead of the 8k that keep the
// provable k-th smallest tight.  (Never more entries than the plain plan: the workspace is sized for that one.)
// `thin` (with `estimate`): 1/64 of the rows -- the scans whose sample is itself expensive (the batched matrix-core scan's
// costs 2.2x a scan of the same rows: 0.58 of 4.9 ms at 512 queries with 1/16); the estimate's rank is then ~48.
BootPlan boot_plan(int64_t R, int64_t Tp, int k, bool halves = false, bool estimate = false, bool thin = false, double margin = 12.0) {
    BootPlan bp{0, 0, 0};
    const int64_t nseg = (Tp + PSH_SEG - 1) / PSH_SEG;
    const int64_t quarter = R / 4;
    int64_t rows = estimate ? (thin ? R / 64 : R / 32) : R / 16;
    // >= 8k segment minima for the provable bound; an estimate wants its rank (~1.5 k x the sampled fraction + 16) well
    // inside the sample: >= 1024 minima and >= 8 x the rank
    int64_t need_rows = (8 * (int64_t)k + nseg - 1) / nseg;
    if (estimate) {
        need_rows = (1024 + nseg - 1) / nseg;
        const double per_row = (double)nseg - margin * (double)k / (double)R;            // entries - 8 x rank (margin 12; 4.5: 3 x rank), per sampled row
        if (per_row > 0.25) { const int64_t nr = (int64_t)(128.0 / per_row) + 1; if (nr > need_rows) need_rows = nr; }
        else need_rows = quarter + 1;                                                    // k too close to the ensemble's size: no sample
    }
    if (rows < need_rows) rows = need_rows;
    if (rows >= 1 && rows <= quarter) {
        if (halves && rows >= 2) { bp.rows = (rows + 1) / 2; bp.per_wave = 2; bp.entries = bp.rows * nseg * 2; return bp; }
        bp.rows = rows; bp.per_wave = 1; bp.entries = rows * nseg; return bp;
    }
    const int64_t lanes_per_row = (Tp + PSH_L - 1) / PSH_L;             // real minima per row
    rows = ((estimate ? 4 : 32) * (int64_t)k + lanes_per_row - 1) / lanes_per_row;
    if (rows > quarter) rows = quarter;
    if (rows < 1 || rows * lanes_per_row < 4 * (int64_t)k) return bp;
    bp.rows = rows; bp.per_wave = 0; bp.entries = rows * nseg * 64;
    return bp;
}
// workspace sizing: the larger of the two per-segment variants (half-segment mode rounds the rows up)
int64_t boot_entries(int64_t R, int64_t Tp, int k) {
    const int64_t a = boot_plan(R, Tp, k, false).entries, b = boot_plan(R, Tp, k, true).entries;
    return a > b ? a : b;
}

// fixed part + cap * 12 bytes per query (the block slices / window slots)
size_t fixed_bytes(int B, int kpad, int64_t min_stride) {
    size_t o = PSH_FUSED_BYTES + PSH_PLAN_BYTES;
    o += align_up(sizeof(QueryState) * (size_t)B, 256);
    o += align_up(sizeof(int) * (size_t)B, 256);
    o += align_up(sizeof(float) * (size_t)B * (size_t)min_stride, 256);
    o += align_up(sizeof(int) * (size_t)B * PSH_MAX_BLOCKS, 256);
    o += align_up(sizeof(int) * PSH_MAX_BLOCKS, 256);
    o += align_up(sizeof(float) * PSH_MAX_BLOCKS, 256);
    o += align_up((size_t)512 * (size_t)((B + 3) & ~3), 256);
    o += align_up(sizeof(int2) * (size_t)B * kpad, 256);
    return o + 1024;  // alignment slack for the four candidate arrays
}

int carve(void* ws, size_t bytes, int B, int k, int64_t min_stride, Workspace* out) {
    const int kpad = next_pow2(k);
    const size_t fixed = fixed_bytes(B, kpad, min_stride);
    if (!ws || bytes <= fixed) return PSH_ERR_WORKSPACE;
    int64_t cap = (int64_t)((bytes - fixed) / (12 * (size_t)B));
    cap &= ~(int64_t)63;
    if (cap > (1 << 30)) cap = 1 << 30;
    if (cap <= 0) return PSH_ERR_WORKSPACE;
    char* p = (char*)ws;
    if (((uintptr_t)p & 255u) != 0) return PSH_ERR_ARG;   // torch allocations are >= 512-byte aligned
    out->fused = (FusedHdr*)p;    p += PSH_FUSED_BYTES;
    out->eplan = (EmbedPlan*)p;   p += PSH_PLAN_BYTES;
    out->qstate = (QueryState*)p; p += align_up(sizeof(QueryState) * (size_t)B, 256);
    out->total = (int*)p;         p += align_up(sizeof(int) * (size_t)B, 256);
    out->minbuf = (float*)p;      p += align_up(sizeof(float) * (size_t)B * (size_t)min_stride, 256);
    out->min_stride = min_stride;
    out->bcount = (int*)p;        p += align_up(sizeof(int) * (size_t)B * PSH_MAX_BLOCKS, 256);
    out->bcount2 = (int*)p;       p += align_up(sizeof(int) * PSH_MAX_BLOCKS, 256);
    out->blockmax = (float*)p;    p += align_up(sizeof(float) * PSH_MAX_BLOCKS, 256);
    out->mq_frag = (void*)p;      p += align_up((size_t)512 * (size_t)((B + 3) & ~3), 256);
    out->sel_rt = (int2*)p;       p += align_up(sizeof(int2) * (size_t)B * kpad, 256);
    out->cand_d = (float*)p;      p += align_up(sizeof(float) * (size_t)B * cap, 256);
    out->cand_rt = (int2*)p;
    out->cap = (int)cap;
    out->kpad = kpad;
    return PSH_OK;
}

// The overlap-friendly launches' lists of admitted windows (psh_stream.hip): 16-byte entries, one compact list per query.
// The workspace's two candidate arrays (4 + 8 bytes per entry of `cap`, adjacent) are taken as ONE region -- up to 65536
// entries per query (the ranking compares every pair: 64 k entries are ~0.3 ms, still a fraction of the separate launches a
// PSH_STATUS_RETRY costs) -- and FusedHdr::cand (16384 entries for the call) serves a workspace too small for more.
#define PSH_STREAM_LIST_MAX 65536
static void stream_cand_list(const Workspace& w, int B, void** list, int* cap_per_query) {
    const int64_t region = (int64_t)((char*)w.cand_rt - (char*)w.cand_d) + (int64_t)sizeof(int2) * B * (int64_t)w.cap;
    int64_t per = region / 16 / B;
    if (per > PSH_STREAM_LIST_MAX) per = PSH_STREAM_LIST_MAX;
    if (per > PSH_STREAM_CAND_CAP / B) { *list = (void*)w.cand_d; *cap_per_query = (int)per; }
    else { *list = (void*)w.fused->cand; *cap_per_query = PSH_STREAM_CAND_CAP / B; }
}

// candidate capacity per query: the scan writes one slice per block (up to
// PSH_MAX_BLOCKS of them), the exhaustive path one slot per window of a row chunk
int recommended_cap(int64_t Tp, int k) {
    const int64_t nseg = (Tp + PSH_SEG - 1) / PSH_SEG;
    int64_t cap = 128 * (int64_t)PSH_MAX_BLOCKS;          // 128 entries per block slice
    if (cap < 64 * (int64_t)k) cap = 64 * (int64_t)k;
    if (cap < k + 4 * nseg * PSH_SEG) cap = k + 4 * nseg * PSH_SEG;   // exhaustive: >= 4 rows per chunk
    return (int)((cap + 63) & ~(int64_t)63);
}

struct Problem {
    int64_t R, T, r_offset, Tp, N;
    int B, W, h, k;
    bool aligned;
    const float* ker;     // embedded scan: emb_d x W kernel matrix (device), else nullptr
    int emb_d;
    int qlen;             // floats per query vector handed over: W, or emb_d (pre-embedded queries)
    bool emb_dense;       // PSH_FLAG_EMBED_DENSE
    bool rows_generic;    // PSH_FLAG_ROWS_GENERIC
    bool emb_taps;        // PSH_FLAG_EMBED_TAPS
    bool emx_split;       // PSH_FLAG_EMBED_MX_SPLIT
    const EmbedPlan* eplan;   // embedded scan, sampled path: the plan region of the workspace when embed_plan_kernel runs, else nullptr
    bool emx;             // PSH_FLAG_EMBED_MX and the kernel fits: embed_mx_kernel (BOOT / FILTER)
};

int check_problem(const float* dataset, int64_t R, int64_t T, int64_t r_offset, const float* queries,
                  int B, int W, int h, int k, float* out_d, int32_t* out_idx, Problem* p,
                  const float* ker = nullptr, int emb_d = 0) {
    if (!dataset || !queries || !out_d || !out_idx) return PSH_ERR_ARG;
    if (emb_d < 0 || (emb_d > 0 && !ker)) return PSH_ERR_ARG;
    if (emb_d > PSH_EMB_MAX_D || (int64_t)emb_d * ((W + 3) & ~3) > PSH_EMB_MAX_TAPS) return PSH_ERR_UNSUPPORTED;
    if (R <= 0 || T <= 0 || B <= 0 || W <= 0 || h < 0 || k <= 0 || r_offset < 0) return PSH_ERR_ARG;
    if (W > PSH_MAX_W || k > PSH_MAX_K || B > PSH_MAX_B_PER_LAUNCH) return PSH_ERR_UNSUPPORTED;
    const int64_t Tp = T - W - h + 1;
    if (Tp <= 0) return PSH_ERR_ARG;
    if (T >= (1ll << 31) - PSH_SEG - 1024 || r_offset + R >= (1ll << 31)) return PSH_ERR_UNSUPPORTED;
    const int64_t N = R * Tp;
    if ((int64_t)k > N) return PSH_ERR_ARG;   // the reference raises too (topk: k out of range)
    p->R = R; p->T = T; p->r_offset = r_offset; p->Tp = Tp; p->N = N;
    p->B = B; p->W = W; p->h = h; p->k = k;
    p->aligned = (((uintptr_t)dataset & 15u) == 0) && (T % 4 == 0);
    p->ker = emb_d > 0 ? ker : nullptr;
    p->emb_d = emb_d;
    p->qlen = emb_d > 0 ? emb_d : W;
    p->emb_dense = false;
    p->rows_generic = false;
    p->emb_taps = false;
    p->emx_split = false;
    p->eplan = nullptr;
    p->emx = false;
    return PSH_OK;
}

struct Plan { int grid; int n_qgroups; int q_per_group; int tile_floats; int wide; };
#define PSH_RESERVED_CUS 4           // PSH_FLAG_RESERVE_CUS: compute units a scan leaves to the side stream (collective, merge)

// Launch-geometry overrides and device-side time stamps for the scripts under tools/: compiled into the
// tuning build only (-DPSH_TUNING, `python -m shadowing_amd._build --tuning`).  The product library reads no
// environment variable and takes no pointer from anywhere but its arguments.
struct Tuning { int mq_i8; int dbg; int px_r1; int wide_min; bool narrow; int bpc; int rows_frac; unsigned long long* dbg_times; unsigned long long* dbg_select; int xcd_skew; int stream_pgrid_per_cu; int stream_skip; int stream_units; int stream_rgrid_per_cu; };
inline Tuning tuning() {
    Tuning t{1, 0, 0, PSH_EMB_WIDE_MIN_B, false, 0, 64, nullptr, nullptr, PSH_FUSED_XCD_SKEW, 2, 0, 2048, 2};
#ifdef PSH_TUNING
    if (const char* e = getenv("PSH_DBG")) t.dbg = atoi(e);
    if (const char* e = getenv("PSH_MQ_I8")) t.mq_i8 = atoi(e);                       // batched scan: 0 = the f16 rejection test (A/B), 2 = the 8-bit one whatever the batch
    if (const char* e = getenv("PSH_PX_R1")) { const int v = atoi(e); if (v >= 2) t.px_r1 = v; }   // prefix-sum scan: rows of the first phase (a large value: one phase)
    if (const char* e = getenv("PSH_EMBED_WIDE_MIN_B")) { const int v = atoi(e); if (v >= 1) t.wide_min = v; }
    t.narrow = getenv("PSH_EMBED_NARROW") != nullptr;
    if (const char* e = getenv("PSH_BLOCKS_PER_CU")) { const int v = atoi(e); if (v > 0) t.bpc = v; }
    if (const char* e = getenv("PSH_ROWS_FRAC")) { const int v = atoi(e); if (v >= 2) t.rows_frac = v; }
    if (const char* e = getenv("PSH_XCD_SKEW")) { const int v = atoi(e); if (v >= -64 && v <= 64) t.xcd_skew = v; }
    if (const char* e = getenv("PSH_STREAM_PGRID")) { const int v = atoi(e); if (v >= 1 && v <= 16) t.stream_pgrid_per_cu = v; }
    if (const char* e = getenv("PSH_STREAM_SKIP")) t.stream_skip = atoi(e);      // bit 0: no sample launch, 1: no ranking, 2: no scan (timing ablations: results are stale)
    if (const char* e = getenv("PSH_STREAM_RGRID")) { const int v = atoi(e); if (v >= 1 && v <= 8) t.stream_rgrid_per_cu = v; }
    if (const char* e = getenv("PSH_STREAM_UNITS")) { const int v = atoi(e); if (v >= 256 && v <= PSH_FUSED_MAX_UNITS) t.stream_units = v; }
    if (const char* e = getenv("PSH_DBG_TIMES_PTR")) t.dbg_times = (unsigned long long*)strtoull(e, nullptr, 0);
    if (const char* e = getenv("PSH_DBG_SELECT_PTR")) t.dbg_select = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    return t;
}
inline int flags_of(const psh_profile* prof) { return prof ? prof->flags : 0; }

int plan_scan(int device, const Problem& p, int64_t n_rows, Plan* plan) {
    const int tile_floats = tile_floats_for(p.W);
    // embedded scan of a batch: 512-thread blocks whose waves carry 12 (suffix rows: 6) queries per evaluation of the
    // embedding (256 VGPRs, one block per CU)
    const Tuning tn = tuning();
    if (p.emx) {
        // embed_mx_kernel: 8 waves (two per SIMD), one block per CU
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        const int nseg = (int)((p.Tp + PSH_SEG - 1) / PSH_SEG);
        const int64_t n_rs = n_rows * nseg;
        const int64_t waves = (int64_t)ncu * 8;
        int n_qgroups = 1;
        if (n_rs < waves && p.B > 1) {
            int64_t g = (waves + n_rs - 1) / n_rs;
            n_qgroups = (int)(g < p.B ? g : p.B);
        }
        const int q_per_group = (p.B + n_qgroups - 1) / n_qgroups;
        n_qgroups = (p.B + q_per_group - 1) / q_per_group;
        const int64_t units = n_rs * n_qgroups;
        if (units >= (1ll << 31)) return PSH_ERR_UNSUPPORTED;
        int64_t grid = (units + 7) / 8;
        if (grid > ncu) grid = ncu;
        if (grid > PSH_MAX_BLOCKS) grid = PSH_MAX_BLOCKS;
        if (grid < 1) grid = 1;
        plan->grid = (int)grid;
        plan->n_qgroups = n_qgroups;
        plan->q_per_group = q_per_group;
        plan->tile_floats = tile_floats;
        plan->wide = 0;
        return PSH_OK;
    }
    bool wide = p.ker && p.B >= tn.wide_min && !tn.narrow;
    // a kernel matrix too large to sit in LDS beside sixteen wave tiles (Foveal(1.15, 0.9, 252): 39 x 252) runs the
    // 8-wave instantiation whatever the batch size
    if (p.ker && !wide && scan_shmem_bytes(tile_floats, p.B, p.emb_d, p.W, PSH_SCAN_THREADS) > PSH_LDS_BYTES) wide = true;
    const int threads = wide ? 512 : PSH_SCAN_THREADS;
    const size_t shmem = scan_shmem_bytes(tile_floats, p.B, p.emb_d, p.W, threads);
    if (shmem > PSH_LDS_BYTES) return PSH_ERR_UNSUPPORTED;
    int bpc = 1, ncu = 0;
    if (!wide) HIP_TRY(scan_blocks_per_cu(p.W, p.aligned, p.ker != nullptr, shmem, &bpc));
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    if (tn.bpc > 0) bpc = tn.bpc;
    if (bpc < 1) bpc = 1;
    if (bpc > 8) bpc = 8;
    const int nseg = (int)((p.Tp + PSH_SEG - 1) / PSH_SEG);
    const int64_t n_rs = n_rows * nseg;
    const int64_t waves = (int64_t)bpc * ncu * (threads / 64);
    // not enough (row, segment) units to fill the chip: also split the queries
    int n_qgroups = 1;
    if (n_rs < waves && p.B > 1) {
        int64_t g = (waves + n_rs - 1) / n_rs;
        n_qgroups = (int)(g < p.B ? g : p.B);
    }
    const int q_per_group = (p.B + n_qgroups - 1) / n_qgroups;
    n_qgroups = (p.B + q_per_group - 1) / q_per_group;
    const int64_t units = n_rs * n_qgroups;
    if (units >= (1ll << 31)) return PSH_ERR_UNSUPPORTED;
    int64_t grid = (units + (threads / 64) - 1) / (threads / 64);
    if (grid > (int64_t)bpc * ncu) grid = (int64_t)bpc * ncu;
    if (grid > PSH_MAX_BLOCKS) grid = PSH_MAX_BLOCKS;
    if (grid < 1) grid = 1;
    plan->grid = (int)grid;
    plan->n_qgroups = n_qgroups;
    plan->q_per_group = q_per_group;
    plan->tile_floats = tile_floats;
    plan->wide = wide ? 1 : 0;
    return PSH_OK;
}

ScanArgs make_scan_args(const float* dataset, const float* queries, const Problem& p, const Workspace& w,
                        const Plan& plan, int64_t row0, int64_t row_stride, int64_t n_rows) {
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.dataset = dataset;
    a.T = p.T;
    a.Tp = (int)p.Tp;
    a.nseg = (int)((p.Tp + PSH_SEG - 1) / PSH_SEG);
    a.W = p.W;
    a.row0 = row0;
    a.row_stride = row_stride;
    a.n_rows = (int)n_rows;
    {   // magic numbers of the unit decode: floor(2^32 / d), see fast_div
        const uint64_t n_rs = (uint64_t)n_rows * (uint64_t)a.nseg;
        a.magic_nrs = n_rs > 1 ? (unsigned)((1ull << 32) / n_rs) : 0u;
        a.magic_nseg = a.nseg > 1 ? (unsigned)((1ull << 32) / (uint64_t)a.nseg) : 0u;
    }
    a.r_offset = p.r_offset;
    a.queries = queries;
    a.ker = p.ker;
    a.hx = p.ker ? queries : nullptr;
    a.emb_d = p.emb_d;
    a.emb_wide = plan.wide;
    a.emb_dense = p.emb_dense ? 1 : 0;                  // PSH_FLAG_EMBED_DENSE: skip the suffix-rows fast path (A/B tests)
    a.emb_mx = p.emx ? (p.emx_split ? 3 : 1) : 0;
    a.emb_taps = p.emb_taps ? 1 : 0;
    a.emb_r1 = tuning().px_r1;
    a.dbg = tuning().dbg;
    a.plan = p.eplan;
    a.B = p.B;
    a.n_qgroups = plan.n_qgroups;
    a.q_per_group = plan.q_per_group;
    a.tile_floats = plan.tile_floats;
    a.k = p.k;
    a.qstate = w.qstate;
    a.minbuf = w.minbuf;
    a.min_stride = w.min_stride;
    a.cand_d = w.cand_d;
    a.cand_rt = w.cand_rt;
    a.bcount = w.bcount;
    a.slice = w.cap / plan.grid;
    a.cap = w.cap;
    return a;
}

// slices = true : rank what the FILTER scan left in `nblk` block slices
// slices = false: rank the first n_fixed flat entries (r < 0 = empty)
SelectArgs make_select_args(const Problem& p, const Workspace& w, float* out_d, int32_t* out_idx, int* status,
                            bool slices, int nblk, int n_fixed) {
    SelectArgs s;
    memset(&s, 0, sizeof(s));
    s.cand_d = w.cand_d;
    s.cand_rt = w.cand_rt;
    s.cand_stride = w.cap;
    s.bcount = slices ? w.bcount : nullptr;
    s.nblk = nblk;
    s.slice = slices ? w.cap / nblk : 0;
    s.total = w.total;
    s.n_fixed = n_fixed;
    s.cap = w.cap;
    s.k = p.k;
    s.kpad = w.kpad;
    s.skip_negative_rows = slices ? 0 : 1;
    s.out_d = out_d;
    s.out_idx = out_idx;
    s.sel_rt = w.sel_rt;
    s.sort_scratch = reinterpret_cast<uint64_t*>(w.cand_rt);   // (r, t) of the selected are in sel_rt by the time the ordering runs
    s.status = status;
    s.qstate = w.qstate;
    return s;
}

int max_total(const Workspace& w, int B, int* out) {
    int cmax = 0;
    for (int b0 = 0; b0 < B; b0 += 256) {
        int tmp[256];
        const int nb = (B - b0) < 256 ? (B - b0) : 256;
        HIP_TRY(hipMemcpy(tmp, w.total + b0, sizeof(int) * nb, hipMemcpyDeviceToHost));
        for (int i = 0; i < nb; ++i) cmax = tmp[i] > cmax ? tmp[i] : cmax;
    }
    *out = cmax;
    return PSH_OK;
}

struct Timer {   // optional per-stage HIP events
    bool on;
    hipStream_t s;
    hipEvent_t ev[8];
    int n = 0;
    Timer(bool enable, hipStream_t stream) : on(enable), s(stream) {}
    int init() {
        if (!on) return PSH_OK;
        for (int i = 0; i < 8; ++i) HIP_TRY(hipEventCreate(&ev[i]));
        return PSH_OK;
    }
    int mark() {
        if (!on) return PSH_OK;
        HIP_TRY(hipEventRecord(ev[n++], s));
        return PSH_OK;
    }
    int elapsed(int i, int j, float* ms) {
        *ms = 0.f;
        if (!on) return PSH_OK;
        HIP_TRY(hipEventElapsedTime(ms, ev[i], ev[j]));
        return PSH_OK;
    }
    ~Timer() { if (on) for (int i = 0; i < 8; ++i) (void)hipEventDestroy(ev[i]); }
};

int run_exhaustive(int device, hipStream_t s, const float* dataset, const float* queries, const float* qnorm,
                   const Problem& p_in, const Workspace& w, float* out_d, int32_t* out_idx, int32_t* out_status,
                   psh_profile* prof) {
    Problem p = p_in;
    p.emx = false;                 // every window is ranked here: the dense chains of embed_scan_kernel, no rejection test
    p.eplan = nullptr;
    const int64_t nseg = (p.Tp + PSH_SEG - 1) / PSH_SEG;
    const bool rows_path = p.Tp == 1 && !p.ker && !p.rows_generic;   // one-window rows: a slot per row (rows_kernel)
    const int64_t slots_per_row = rows_path ? 1 : nseg * PSH_SEG;
    if ((int64_t)w.cap < (int64_t)p.k + slots_per_row) return PSH_ERR_WORKSPACE;
    const bool stages = prof && prof->mode == PSH_PROFILE_STAGES;
    const bool events = prof && prof->mode == PSH_PROFILE_EVENTS && prof->ev_scan_begin && prof->ev_scan_end;
    Timer tm(stages, s);
    int rc = tm.init(); if (rc) return rc;
    rc = tm.mark(); if (rc) return rc;
    PrepArgs pa{queries, qnorm, p.B, p.qlen, w.qstate, w.total, out_status};
    HIP_TRY(launch_prep(pa, s));
    rc = tm.mark(); if (rc) return rc;
    if (events) HIP_TRY(hipEventRecord((hipEvent_t)prof->ev_scan_begin, s));
    int64_t rows_per_chunk = ((int64_t)w.cap - p.k) / slots_per_row;
    if (rows_per_chunk > p.R) rows_per_chunk = p.R;
    int grid_used = 0;
    for (int64_t r0 = 0; r0 < p.R; r0 += rows_per_chunk) {
        const int64_t nr = (r0 + rows_per_chunk <= p.R) ? rows_per_chunk : (p.R - r0);
        Plan plan;
        rc = plan_scan(device, p, nr, &plan); if (rc) return rc;
        grid_used = plan.grid;
        const int n_slots = (int)(nr * slots_per_row);
        // the running best (empty on the first chunk) sits right behind the window slots
        ReseedArgs ra{out_d, out_idx, w.qstate, w.cand_d, w.cand_rt, (int64_t)w.cap, n_slots, p.k};
        HIP_TRY(launch_reseed(ra, p.B, s));
        ScanArgs sa = make_scan_args(dataset, queries, p, w, plan, r0, 1, nr);
        if (rows_path) {
            int ncu = 0;
            HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
            int64_t gb = (nr + 127) / 128;
            if (gb > 8 * (int64_t)ncu) gb = 8 * (int64_t)ncu;
            HIP_TRY(launch_rows(sa, PSH_MODE_ALL, (int)(gb < 1 ? 1 : gb), s));
        } else {
            HIP_TRY(launch_scan(sa, PSH_MODE_ALL, p.aligned, plan.grid, s));
        }
        SelectArgs se = make_select_args(p, w, out_d, out_idx, nullptr, false, 0, n_slots + p.k);
        HIP_TRY(launch_select(se, p.B, s));
    }
    rc = tm.mark(); if (rc) return rc;
    if (events) HIP_TRY(hipEventRecord((hipEvent_t)prof->ev_scan_end, s));
    if (prof) { prof->path = 1; prof->grid_blocks = grid_used; prof->n_sample_rows = 0; }
    if (stages) {
        HIP_TRY(hipStreamSynchronize(s));
        tm.elapsed(0, 1, &prof->prep_ms);
        tm.elapsed(1, 2, &prof->scan_ms);
        tm.elapsed(0, 2, &prof->total_ms);
        prof->sample_ms = prof->threshold_ms = prof->select_ms = 0.f;
        rc = max_total(w, p.B, &prof->n_candidates); if (rc) return rc;
    }
    return PSH_OK;
}

}  // namespace

extern "C" {

int psh_version(void) { return PSH_VERSION; }

const char* psh_strerror(int code) {
    switch (code) {
        case PSH_OK: return "ok";
        case PSH_ERR_ARG: return "invalid argument (null pointer, non-positive size, k larger than the number of windows, misaligned workspace)";
        case PSH_ERR_UNSUPPORTED: return "unsupported size (W > PSH_MAX_W, k > PSH_MAX_K, or int32 index overflow)";
        case PSH_ERR_WORKSPACE: return "workspace too small (see psh_workspace_bytes)";
        case PSH_ERR_HIP: return "HIP runtime error (see psh_last_hip_error)";
        case PSH_ERR_COMM: return "RCCL error (see psh_last_comm_error)";
        default: return "unknown error";
    }
}

const char* psh_last_hip_error(void) { return g_hip_err; }

int psh_workspace_bytes(int64_t R, int64_t T, int B, int W, int h, int k, size_t* out_bytes) {
    if (!out_bytes || R <= 0 || T <= 0 || B <= 0 || W <= 0 || h < 0 || k <= 0) return PSH_ERR_ARG;
    if (W > PSH_MAX_W || k > PSH_MAX_K) return PSH_ERR_UNSUPPORTED;
    const int64_t Tp = T - W - h + 1;
    if (Tp <= 0) return PSH_ERR_ARG;
    const int cap = recommended_cap(Tp, k);
    *out_bytes = fixed_bytes(B, next_pow2(k), boot_entries(R, Tp, k)) + (size_t)12 * (size_t)B * (size_t)cap + 64 * 12 * (size_t)B;
    return PSH_OK;
}

int psh_stream_create_reserving(int device, int reserve_cus, void** out_stream, int* out_reserved) {
    if (!out_stream || reserve_cus < 0) return PSH_ERR_ARG;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    int ncu = 0;
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    if (reserve_cus * 4 > ncu) reserve_cus = 0;             // a small device: nothing is reserved
    // one bit per compute unit; the FIRST `reserve_cus` bits are cleared (on a multi-XCD part the bits interleave over
    // the XCDs, so eight of them are one compute unit per XCD; were they not, they would be eight units of one XCD --
    // either way that many units stay free)
    uint32_t mask[32];
    const int words = (ncu + 31) / 32;
    if (words > 32) return PSH_ERR_UNSUPPORTED;
    for (int w = 0; w < words; ++w) {
        uint32_t m = 0xffffffffu;
        const int rem = ncu - 32 * w;
        if (rem < 32) m = (rem <= 0) ? 0u : ((1u << rem) - 1u);
        mask[w] = m;
    }
    for (int b = 0; b < reserve_cus; ++b) mask[b / 32] &= ~(1u << (b % 32));
    hipStream_t s = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
    *out_stream = (void*)s;
    if (out_reserved) *out_reserved = reserve_cus;
    return PSH_OK;
}

int psh_stream_destroy(int device, void* stream) {
    if (!stream) return PSH_ERR_ARG;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return PSH_OK;
}

int psh_workspace_init(int device, void* stream, void* workspace, size_t workspace_bytes) {
    if (!workspace || ((uintptr_t)workspace & 255u) != 0) return PSH_ERR_ARG;
    if (workspace_bytes < PSH_FUSED_BYTES) return PSH_ERR_WORKSPACE;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    HIP_TRY(launch_fused_init((FusedHdr*)workspace, (hipStream_t)stream));
    return PSH_OK;
}

int psh_query_norm(int device, void* stream, const float* queries, int B, int W, float* out_qnorm) {
    if (!queries || !out_qnorm || B <= 0 || W <= 0) return PSH_ERR_ARG;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    HIP_TRY(launch_qnorm(queries, B, W, out_qnorm, (hipStream_t)stream));
    return PSH_OK;
}

static int scan_exhaustive_impl(int device, void* stream, const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                                const float* queries, const float* qnorm, int B, int W, int h, int k,
                                const float* ker, int emb_d,
                                float* out_d, int32_t* out_idx, int32_t* out_status,
                                void* workspace, size_t workspace_bytes, psh_profile* profile) {
    Problem p;
    int rc = check_problem(dataset, R, T, r_offset, queries, B, W, h, k, out_d, out_idx, &p, ker, emb_d);
    if (rc) return rc;
    p.emb_dense = (flags_of(profile) & PSH_FLAG_EMBED_DENSE) != 0;
    p.rows_generic = (flags_of(profile) & PSH_FLAG_ROWS_GENERIC) != 0;
    Workspace w;
    rc = carve(workspace, workspace_bytes, B, k, boot_entries(p.R, p.Tp, k), &w);
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    return run_exhaustive(device, (hipStream_t)stream, dataset, queries, qnorm, p, w, out_d, out_idx, out_status, profile);
}

int psh_scan_topk_exhaustive(int device, void* stream, const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                             const float* queries, const float* qnorm, int B, int W, int h, int k,
                             float* out_d, int32_t* out_idx, int32_t* out_status,
                             void* workspace, size_t workspace_bytes, psh_profile* profile) {
    return scan_exhaustive_impl(device, stream, dataset, R, T, r_offset, queries, qnorm, B, W, h, k, nullptr, 0,
                                out_d, out_idx, out_status, workspace, workspace_bytes, profile);
}

int psh_scan_topk_embedded_exhaustive(int device, void* stream, const float* dataset, int64_t R, int64_t T,
                                      int64_t r_offset, const float* kernel, int d, int K,
                                      const float* hx, const float* hxnorm, int B, int h, int k,
                                      float* out_d, int32_t* out_idx, int32_t* out_status,
                                      void* workspace, size_t workspace_bytes, psh_profile* profile) {
    if (!kernel || d <= 0) return PSH_ERR_ARG;
    if (T == (int64_t)K + h) return PSH_ERR_UNSUPPORTED;     // one-window rows: psh_embed_rows + psh_scan_topk (see psh.h)
    return scan_exhaustive_impl(device, stream, dataset, R, T, r_offset, hx, hxnorm, B, K, h, k, kernel, d,
                                out_d, out_idx, out_status, workspace, workspace_bytes, profile);
}

// queries one step of the long-window scan serves (34 <= W <= 256): up to three -- as many as put their fragment tables in LDS beside
// the waves' rows (W <= ~230: three; up to 256: two); 0 = not a long window
static int long_queries_per_step(int W) {
    if (!stream_long_supported(W)) return 0;
    for (int nq = PSH_STREAM_MAX_Q; nq > 1; --nq)
        if (stream_scan_long_shmem_bytes(W, nq) <= PSH_LDS_BYTES) return nq;
    return 1;
}

// what psh_shadow_blocking adds to a one-query call: the fused launch gathers the winners' paths itself and sets completion
// words the host polls; `taken` says whether the call was served that way (else: the caller enqueues the gather and waits
// for the stream)
struct BlockingExtras {
    const float* g_ds; float* g_out; int64_t g_T; int g_C, g_len;
    unsigned* done; unsigned done_val;
    bool taken; int shards;
    const float* q_host; const float* hint_host;      // the query (W floats) and, nullable, the admission level as the HOST reads them
};

static int scan_topk_impl(int device, void* stream, const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                          const float* queries, const float* qnorm, int B, int W, int h, int k,
                          const float* ker, int emb_d,
                          float* out_d, int32_t* out_idx, int32_t* out_status,
                          void* workspace, size_t workspace_bytes, psh_profile* profile, BlockingExtras* bx = nullptr) {
    Problem p;
    int rc = check_problem(dataset, R, T, r_offset, queries, B, W, h, k, out_d, out_idx, &p, ker, emb_d);
    if (rc) return rc;
    if (!out_status) return PSH_ERR_ARG;
    // Several queries with a window the batched kernels' bands do not reach (they stop at W = 25): a LOOP of the steps that do
    // have a matrix-core rejection test, inside the call --
    //   34 <= W <= 256: up to three queries per step (the three launches with the long-window scan, psh_stream.hip: the
    //                   queries share the pass, the conversion and the energies' MFMAs);
    //   26 <= W <= 33, four queries and more: three queries per step (the three launches' 2-3 query form).
    // Round 6: what is left to the loop is PSH_FLAG_LONG_LOOP and shapes the batched long-window scan does not take (use_lq below
    // serves four queries and more with 26 <= W <= 256 and two / three that do not ride one pass).
    // The one-pass vector-ALU filter costs W fma per window and query: R = 32768, T = 4096, W = 126 -- 2 / 4 / 16 / 64 queries
    // 1.14 / 2.19 / 8.6 / 33.6 ms in one pass, 0.34 / 0.65 / 2.6 / 10.6 ms as a loop; W = 252: 2.2 .. 67 against 0.50 .. 16.2 ms
    // (tools/long_batch_probe.py).  Status words stay per query; a RETRY of any step sends the caller's WHOLE call to
    // PSH_FLAG_NO_FUSE, as the protocol says.
    const int long_q = long_queries_per_step(W);
    // four queries and more with a long window -- or two / three that do not ride one pass of the three launches (W > 97 / 145:
    // their tables do not fit beside the scan's rows) --: ONE pass per chunk of queries of the batched long-window scan (psh_lq.hip)
    // through the separate launches' pipeline, instead of the loop of steps below
    const bool use_lq = !ker && p.Tp > 1 && (B >= 4 || (long_q > 0 && B > long_q)) && scan_lq_supported(W, B, T) &&
                        !(flags_of(profile) & (PSH_FLAG_FILTER_VALU | PSH_FLAG_NO_FUSE | PSH_FLAG_LONG_LOOP));
    const int per_step = !ker && p.Tp > 1 ? (long_q > 0 && B > long_q ? long_q : (W >= 26 && W <= 33 && B > PSH_STREAM_MAX_Q ? PSH_STREAM_MAX_Q : 0)) : 0;
    // (the loop pays only when its sub-calls get the three launches: 5 k candidates in a query's list of 65536, a sample of 256
    //  units and more -- a call outside that would be B / 3 passes with the vector-ALU filter instead of one; psh_profile then
    //  describes the LAST step of the loop: path, grid, and in PSH_PROFILE_EVENTS mode the bracket of that step's scan)
    const bool step_fits = 5 * (int64_t)k <= 65536 && p.R * ((p.Tp + PSH_SEG - 1) / PSH_SEG) >= 1024;
    if (per_step && step_fits && !use_lq && !(profile && profile->mode == PSH_PROFILE_STAGES) && !(flags_of(profile) & (PSH_FLAG_FILTER_VALU | PSH_FLAG_NO_FUSE))) {
        psh_profile sub;
        for (int b = 0; b < B; b += per_step) {
            const int nb = B - b < per_step ? B - b : per_step;
            if (profile) { sub = *profile; if (profile->tau_hint) sub.tau_hint = profile->tau_hint + b; }
            rc = scan_topk_impl(device, stream, dataset, R, T, r_offset, queries + (size_t)b * W, qnorm ? qnorm + b : nullptr, nb, W, h, k,
                                nullptr, 0, out_d + (size_t)b * k, out_idx + (size_t)b * k * 2, out_status + b, workspace, workspace_bytes,
                                profile ? &sub : nullptr);
            if (rc) return rc;
        }
        if (profile) { const float* hint0 = profile->tau_hint; *profile = sub; profile->tau_hint = hint0; }
        return PSH_OK;
    }
    // A dense embedding's batch beyond what the matrix-core scan takes in one call (its per-query pass runs on the matrix cores
    // for up to 256 queries -- PSH_EMX_QM_MAX_B -- and the queries' constants sit in LDS): chunks of the largest supported size inside the call instead of the vector-ALU
    // scan for all of them (configs[4] with 512 query dates: 46 -> 10 ms per GPU).  Status words stay per query.
    if (ker && (flags_of(profile) & PSH_FLAG_EMBED_MX) && !(flags_of(profile) & PSH_FLAG_EMBED_DENSE) && p.Tp > 1 && B > 3 &&
        (B > 256 || !embed_mx_supported(emb_d, W, B, tile_floats_for(W))) && !(profile && profile->mode == PSH_PROFILE_STAGES)) {
        int chunk = 0;
        for (int c = B < 256 ? B : 256; c >= 3; c = c > 32 ? c - 32 : c - 1)
            if (embed_mx_supported(emb_d, W, c, tile_floats_for(W))) { chunk = c; break; }
        if (chunk >= 3) {
            const int n_chunks = (B + chunk - 1) / chunk;
            const int per = (B + n_chunks - 1) / n_chunks;                    // even chunks
            psh_profile sub;
            for (int b = 0; b < B; b += per) {
                const int nb = B - b < per ? B - b : per;
                if (profile) { sub = *profile; if (profile->tau_hint) sub.tau_hint = profile->tau_hint + b; }
                rc = scan_topk_impl(device, stream, dataset, R, T, r_offset, queries + (size_t)b * emb_d, qnorm ? qnorm + b : nullptr, nb, W, h, k,
                                    ker, emb_d, out_d + (size_t)b * k, out_idx + (size_t)b * k * 2, out_status + b, workspace, workspace_bytes,
                                    profile ? &sub : nullptr);
                if (rc) return rc;
            }
            if (profile) { const float* hint0 = profile->tau_hint; *profile = sub; profile->tau_hint = hint0; }
            return PSH_OK;
        }
    }
    p.emb_dense = (flags_of(profile) & PSH_FLAG_EMBED_DENSE) != 0;
    p.rows_generic = (flags_of(profile) & PSH_FLAG_ROWS_GENERIC) != 0;
    p.emb_taps = (flags_of(profile) & PSH_FLAG_EMBED_TAPS) != 0;
    p.emx_split = (flags_of(profile) & PSH_FLAG_EMBED_MX_SPLIT) != 0;
    p.emx = p.ker && (flags_of(profile) & PSH_FLAG_EMBED_MX) && !p.emb_dense && p.Tp > 1 &&
            embed_mx_supported(p.emb_d, p.W, p.B, tile_floats_for(p.W));
    Workspace w;
    rc = carve(workspace, workspace_bytes, B, k, boot_entries(p.R, p.Tp, k), &w);
    if (rc) return rc;
    if ((int64_t)w.cap < (int64_t)k + ((p.Tp + PSH_SEG - 1) / PSH_SEG) * PSH_SEG) return PSH_ERR_WORKSPACE;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    hipStream_t s = (hipStream_t)stream;
    // a linear embedding that may be Foveal-like on one interval: the matrix is looked at on the device (one small launch),
    // and of the two kernels launched per stage the one the structure belongs to does the work (psh_embed_px.hip)
    const bool want_plan = p.ker && !p.emb_dense && !p.emb_taps && !p.emx && p.Tp > 1 &&
                           embed_px_supported(tile_floats_for(p.W), p.B, p.emb_d, p.W, false) &&
                           embed_px_supported(tile_floats_for(p.W), p.B, p.emb_d, p.W, true);
    // (in front of the decision between the sampled and the exhaustive path: with PSH_FLAG_EMBED_PLAN_KEEP the next call may
    //  take the other one)
    if (want_plan && !(flags_of(profile) & PSH_FLAG_EMBED_PLAN_KEEP)) HIP_TRY(launch_embed_plan(p.ker, p.emb_d, p.W, w.eplan, s));

    // small problems (everything fits the candidate buffer), one-window rows (their
    // numerator uses another reduction order, handled by the exhaustive kernel only) or
    // a sample too thin to be useful: exhaustive path
    // the cheap test of the full scan runs on the matrix cores where that is implemented
    // (PSH_FLAG_FILTER_VALU keeps it on the vector ALUs: comparison runs, tools/)
    bool use_mx = !p.ker && scan_mx_supported(p.W, p.B);
    bool use_mq = !p.ker && scan_mq_supported(p.W, p.B);      // batched queries: 4 queries x 8 shifts per MFMA
    if (flags_of(profile) & PSH_FLAG_FILTER_VALU) use_mx = use_mq = false;
    // the batched scan's rejection test: the 8-bit product (scan_mq8_kernel) unless the caller asks for f16
    // (from 32 queries on: a segment's set-up is ~1.5 k cycles dearer with the 8-bit test -- two passes over the staged values, the
    //  energies turned into integers, the levels of every query -- and that is what a small batch pays for; 4 queries 230 against
    //  179 us per call, 16: 303 / 279, 32: 373 / 395, 128: 778 / 1083)
    const bool mq_i8 = tuning().mq_i8 != 0 && !(flags_of(profile) & PSH_FLAG_MQ_F16) && (p.B >= 32 || tuning().mq_i8 > 1);
    // (half-segment mode measured for the single-query scan: bootstrap 17.4 -> 12.3 us, but tau admits twice as
    // much and the scan's exact rechecks cost 4.3 us more -- 135.7 vs 134.3 us per step; left off)
    // (the matrix-core embedded scan samples one minimum per HALF segment; a sample too thin for that plan is taken by
    // embed_scan_kernel instead -- the full scan still runs on the matrix cores)
    BootPlan bp = boot_plan(p.R, p.Tp, k, p.emx, p.ker != nullptr || !use_mx, !p.ker && !use_mx);
    // A single query with a LARGE k (the reference's own example call: Identity(20), R = 32768, k = 8192 -- testing.ipynb): the
    // provable bound wants 8 k segment minima, more than a quarter of the rows have; the plan then fell to one minimum per LANE
    // (264 k minima, 0.19 ms to find the k-th among them, a bound that admitted 4.6 k candidates per k).  Such a call admits below
    // an ESTIMATE instead, like the embedded scans do: 1/32 of the rows, the r2-th smallest of their segment minima with ~1.5 k
    // windows of the ensemble expected below it; fewer than k found -> status -> the caller's exhaustive pass.
    bool mx_estimate = false;
    // (a sample of moderate size keeps its provable bound: per lane up to 100 k minima, per segment up to an eighth of the rows)
    if (use_mx && ((bp.per_wave == 0 && bp.entries > 100000) || (bp.per_wave == 1 && bp.rows * 8 > p.R))) {
        const BootPlan be = boot_plan(p.R, p.Tp, k, false, true, false, 4.5);
        if (be.per_wave == 1 && be.entries <= w.min_stride) { bp = be; mx_estimate = true; }
    }
    const bool boot_emx = p.emx && bp.per_wave == 2;
    if (p.emx && !boot_emx) bp = boot_plan(p.R, p.Tp, k, false, true);
    // one-window rows (T == W + h; PathDistance.forward_topk's N pre-embedded points): rows_kernel, a row per lane.
    // Its bootstrap takes one exact value per sampled row.
    const bool rows_path = p.Tp == 1 && !p.ker && !p.rows_generic;
    bool rows_wave_min = false;
    if (rows_path) {
        const int frac = tuning().rows_frac;
        int64_t ns = p.R / frac > 16 * (int64_t)k ? p.R / frac : 16 * (int64_t)k;
        if (ns > p.R / 2) ns = p.R / 2;
        if (ns > w.min_stride) ns = w.min_stride;
        bp.rows = ns >= 2 * (int64_t)k ? ns : 0;
        bp.per_wave = 1;
        bp.entries = bp.rows;
        // the estimate's rank (below) against the chunks of 64 rows of the sample: sparse enough -> one minimum per chunk
        const int64_t r2r = (6 * (int64_t)k * bp.rows + 2 * p.R - 1) / (2 * p.R) + 16, chunks = (bp.rows + 63) / 64;
        rows_wave_min = bp.rows > 0 && r2r < k && 8 * r2r <= chunks;
        if (rows_wave_min) bp.entries = chunks;
        use_mx = use_mq = false;
    }
    // the caller's admission levels (psh_profile.tau_hint): no bootstrap sample anywhere below
    const float* hint = profile ? profile->tau_hint : nullptr;
    const int64_t n_sample = bp.rows;
    if (p.R * ((p.Tp + PSH_SEG - 1) / PSH_SEG) * PSH_SEG + k <= (int64_t)w.cap || (p.Tp == 1 && !rows_path) || (n_sample == 0 && !hint))
        return run_exhaustive(device, s, dataset, queries, qnorm, p, w, out_d, out_idx, out_status, profile);
    const int64_t stride = n_sample > 0 ? p.R / n_sample : 1;
    const int64_t row0 = stride / 2;
    if (want_plan) p.eplan = w.eplan;

    const bool stages = profile && profile->mode == PSH_PROFILE_STAGES;
    const bool events = profile && profile->mode == PSH_PROFILE_EVENTS && profile->ev_scan_begin && profile->ev_scan_end;

    // ---- the step as three overlap-friendly launches (psh_stream.hip: sample + levels in one-wave blocks, the barrier-free
    // scan, the ranking): ONE query when the caller says other steps are in flight on other streams (PSH_FLAG_OVERLAP), and
    // always for two or three queries -- they ride one pass over the ensemble at about one query's cost, where the batched
    // scan (sized for hundreds of queries) streams at half rate.
    {
        const bool small_batch = !p.ker && !rows_path && B >= 2 && B <= PSH_STREAM_MAX_Q && !(flags_of(profile) & PSH_FLAG_FILTER_VALU);
        const bool one_overlap = use_mx && B == 1 && (flags_of(profile) & PSH_FLAG_OVERLAP);
        // ONE query with a long window (34 <= W <= 256): the same three launches, flag or no flag, with the scan's banded product
        // as a K-loop (stream_scan_long_kernel) -- otherwise such a call has only the vector-ALU filter of scan_kernel
        // (the long-window scan's deferred survivors pack t into 30 bits)
        const bool one_long = !p.ker && !rows_path && long_q > 0 && B <= long_q && !(flags_of(profile) & PSH_FLAG_FILTER_VALU) && p.T < (1ll << 30);
        if ((small_batch || one_overlap || one_long) && !rows_path && !stages && !(flags_of(profile) & PSH_FLAG_NO_FUSE) &&
            (scan_fused_supported(p.W) || one_long)) {
            int ncu = 0;
            HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
            const Tuning tn = tuning();
            const int64_t nseg = (p.Tp + PSH_SEG - 1) / PSH_SEG;
            // a thinner sample than the fused launch's: its megabytes are HBM traffic beside ANOTHER step's scan here.  2048
            // units (what the exchange area holds, split between the queries); the level is the r2p-th smallest minimum (below): the
            // k best windows of the ensemble put kf = k x sampled fraction = 16 expected minima below their level at the benchmark's
            // sizes, P(Poisson(16) >= 40) = 3e-7 that fewer than k windows lie below the estimate (-> PSH_STATUS_RETRY)
            int64_t units_cap = tn.stream_units < 2048 ? tn.stream_units : 2048;     // (the sample kernel keeps a query's minima in registers: <= 2048)
            // (a long window's exact sample chains cost W / 20 of the benchmark's: half the units -- the level's rank stays at its
            //  floor of 24 with 8 minima expected below the k-th distance)
            // (round 6: without a hint a long window's sample runs on the matrix cores -- stream_sample_long_kernel, upper bounds of the
            //  minima -- and takes the full 2048 units; PSH_STREAM_SKIP bit 3 of the tuning build: the exact chains, for A/B runs)
            const bool long_mx_sample = one_long && !hint && !(tn.stream_skip & 8) &&
                                        stream_sample_long_shmem_bytes(p.W, B) <= (size_t)PSH_LDS_BYTES;
            if (one_long && !long_mx_sample && units_cap > 1024) units_cap = 1024;
            if (units_cap > PSH_FUSED_MAX_UNITS / B) units_cap = PSH_FUSED_MAX_UNITS / B;
            int64_t rows_p = units_cap / nseg;
            if (rows_p > p.R / 4) rows_p = p.R / 4;
            if (rows_p < 1) rows_p = 1;
            const int64_t units_p = rows_p * nseg;
            // the level's rank among the sampled minima: the k best windows of the ensemble put at most kf = k x sampled fraction
            // minima below their level (Poisson), so the (kf + 5 sqrt(kf) + 4)-th smallest lies above it but for five sigma -- 40 at
            // the benchmark's kf = 16 (what 2 kf + 8 gave until round 6: the same there, but twelve sigma at kf = 128, where it
            // admitted twice the windows a smooth ensemble's lists hold)
            const double kf = (double)k * (double)rows_p / (double)p.R;
            int64_t r2p = (int64_t)ceil(kf + 5.0 * sqrt(kf)) + 4;
            if (r2p < 24) r2p = 24;
            int64_t grid_p = (int64_t)tn.stream_pgrid_per_cu * ncu;
            // (a long window's sample is its exact chains -- 1024 windows x W taps per unit, 11 us at W = 126 -- not its bytes:
            //  one unit per wave instead of two, unless the caller says other steps' scans share the chip: three streams measured
            //  +2 % with it, a lone stream's call 3 - 10 % less -- 3 queries with W = 126: 317 -> 284 us)
            if (one_long && !(flags_of(profile) & PSH_FLAG_OVERLAP)) grid_p *= 2;
            if (grid_p > units_p) grid_p = units_p;
            int64_t grid_s = ncu;
            // a stream made by psh_stream_create_reserving: one block per compute unit the stream may use
            if ((flags_of(profile) & PSH_FLAG_RESERVE_CUS) && ncu >= 4 * PSH_STREAM_RESERVED_CUS) grid_s = ncu - PSH_STREAM_RESERVED_CUS;
            const int64_t n_rs = p.R * nseg;
            if (grid_s * (PSH_SCAN_THREADS / 64) > n_rs) grid_s = (n_rs + (PSH_SCAN_THREADS / 64) - 1) / (PSH_SCAN_THREADS / 64);
            const int front = B == 1 ? PSH_FUSED_FRONT : 2 * PSH_FUSED_FRONT;
            int cand_cap = 0;
            void* cand_list = nullptr;
            stream_cand_list(w, B, &cand_list, &cand_cap);
            const int logical = PSH_SEG + p.W + 3;
            const int tile_fl = (logical + ((logical >> 6) << 2) + 4 + 3) & ~3;
            int tb = 0;
            while ((1ll << tb) < p.Tp) ++tb;
            if ((one_long ? stream_scan_long_shmem_bytes(p.W, B) : stream_scan_shmem_bytes_q(tile_fl, B)) <= PSH_LDS_BYTES &&
                ((units_p >= 256 && r2p <= units_p / 2) || hint) &&
                5 * (int64_t)k <= (int64_t)cand_cap && 5 * (int64_t)k * B <= grid_s * front * 2) {
                Plan plan_s{(int)grid_s, 1, B, tile_fl, 0};
                ScanArgs fa = make_scan_args(dataset, queries, p, w, plan_s, 0, 1, p.R);
                fa.tile_floats = tile_fl;
                fa.dbg_times = tn.dbg_times;
                FusedArgs fu;
                memset(&fu, 0, sizeof(fu));
                fu.hdr = w.fused;
                fu.boot_units = (int)units_p;
                fu.boot_row_stride = p.R / rows_p;
                fu.boot_row0 = fu.boot_row_stride / 2;
                fu.rank = (int)r2p;
                fu.qnorm_in = qnorm;
                fu.out_d = out_d;
                fu.out_idx = out_idx;
                fu.status = out_status;
                fu.total = w.total;
                fu.tbits = ((p.R + p.r_offset) <= (1ll << (32 - tb))) ? tb : -1;   // rows r_offset .. r_offset + R - 1, t < Tp
                fu.front = front;
                fu.nq = B;
                fu.units_stride = (int)((PSH_FUSED_MAX_UNITS / B) & ~3);
                fu.cand_cap = cand_cap;
                fu.cand_list = cand_list;
                fu.k_out = k;
                fu.tau_hint = hint;
                if (hint) grid_p = 1;                  // nothing is sampled: one block derives scale, thresholds and the fragment table from the hints
                if (!(tn.stream_skip & 1)) HIP_TRY(long_mx_sample ? launch_stream_sample_long(fa, fu, (int)grid_p, s)
                                                                  : launch_stream_sample(fa, fu, p.aligned, (int)grid_p, tile_floats_for(p.W), s));
                if (events) HIP_TRY(hipEventRecord((hipEvent_t)profile->ev_scan_begin, s));
                if (!(tn.stream_skip & 4)) HIP_TRY(one_long ? launch_stream_scan_long(fa, fu, p.aligned, (int)grid_s, s)
                                                            : launch_stream_scan(fa, fu, p.aligned, (int)grid_s, s));
                if (events) HIP_TRY(hipEventRecord((hipEvent_t)profile->ev_scan_end, s));
                // (~2.5 k candidates per query: <= 8 own ones per wave, ONE pass over all of them)
                if (!(tn.stream_skip & 2)) HIP_TRY(launch_stream_rank(fa, fu, tn.stream_rgrid_per_cu * ncu, s));
                if (profile) { profile->path = 3; profile->n_sample_rows = (int)rows_p; profile->grid_blocks = (int)grid_s; }
                return PSH_OK;
            }
        }
    }

    // ---- the whole step as ONE launch (psh_fused.hip): a single query on the matrix-core scan whose bootstrap sample is
    // one minimum per (row, segment) unit and fits the exchange area.  Anything that goes wrong inside raises
    // PSH_STATUS_RETRY in out_status and the caller reruns with PSH_FLAG_NO_FUSE.
    if (use_mx && !rows_path && !stages && !(flags_of(profile) & PSH_FLAG_NO_FUSE) && scan_fused_supported(p.W)) {
        Plan plan_f;
        rc = plan_scan(device, p, p.R, &plan_f); if (rc) return rc;
        if ((flags_of(profile) & PSH_FLAG_RESERVE_CUS) && plan_f.grid > 2 * PSH_RESERVED_CUS) plan_f.grid -= PSH_RESERVED_CUS;
        // its own sample: only an ESTIMATE of the k-th smallest acc is needed (what is admitted is verified exactly, and
        // "at least k admitted" proves the result complete), so at most PSH_FUSED_MAX_UNITS (row, segment) units -- one
        // minimum each -- spread over at most a quarter of the ensemble
        const int64_t nseg = (p.Tp + PSH_SEG - 1) / PSH_SEG;
        int64_t rows_f = PSH_FUSED_MAX_UNITS / nseg;
        if (rows_f > p.R / 4) rows_f = p.R / 4;
        const int64_t units_f = rows_f * nseg;
        // the admission level: the rank-th smallest sampled minimum -- ~2k windows of the whole ensemble expected below it
        // (~3k when that rank is small and therefore noisy)
        int64_t r2 = rows_f > 0 ? (2 * (int64_t)k * rows_f + p.R - 1) / p.R : 0;
        r2 += r2 < 64 ? r2 / 2 + 16 : 8;
        // (~2.5 k candidates are expected: they must fit the blocks' front lists with room to spare)
        if (plan_f.grid <= PSH_FUSED_MAX_BLOCKS && ((rows_f >= 1 && units_f >= 256 && r2 <= units_f / 2) || hint) &&
            5 * (int64_t)k <= (int64_t)plan_f.grid * PSH_FUSED_FRONT) {
            if (rows_f < 1) rows_f = 1;
            const int64_t stride_f = p.R / rows_f, row0_f = stride_f / 2;
            ScanArgs fa = make_scan_args(dataset, queries, p, w, plan_f, 0, 1, p.R);
            const int logical = PSH_SEG + p.W + 3;
            fa.tile_floats = (logical + ((logical >> 6) << 2) + 4 + 3) & ~3;
            if (scan_fused_shmem_bytes(fa.tile_floats) <= PSH_LDS_BYTES) {
                FusedArgs fu;
                memset(&fu, 0, sizeof(fu));
                fu.hdr = w.fused;
                fu.boot_units = (int)units_f;
                fu.boot_row0 = row0_f;
                fu.boot_row_stride = stride_f;
                fu.rank = (int)r2;
                fu.qnorm_in = qnorm;
                fu.out_d = out_d;
                fu.out_idx = out_idx;
                fu.status = out_status;
                fu.total = w.total;
                fu.tau_hint = hint;
                // give-up time of a poll at the 100 MHz wall clock: 2 ms (a block that is not resident); 20 ms when a
                // collective shares the chip (its workgroups may hold a few CUs until the peers arrive)
                fu.spin_ticks = (flags_of(profile) & PSH_FLAG_RESERVE_CUS) ? 2000000 : 200000;
                {
                    int tb = 0;
                    while ((1ll << tb) < p.Tp) ++tb;
                    fu.tbits = ((p.R + p.r_offset) <= (1ll << (32 - tb))) ? tb : -1;   // rows r_offset .. r_offset + R - 1, t < Tp
                }
                fu.xcd_skew = (plan_f.grid % 8 == 0) ? tuning().xcd_skew : 0;      // (a grid that is not whole rounds of the 8 XCDs: no assumption)
                fa.dbg_times = tuning().dbg_times;
                // (a caller that asked for the overlap-friendly launches may be on a stream that cannot hold the fused
                //  launch's blocks all at once -- a CU mask, other scans in flight: where they do not apply, the separate
                //  launches serve the call, never the fused one)
                if (!(flags_of(profile) & PSH_FLAG_OVERLAP)) {
                if (bx) {
                    fu.g_ds = bx->g_ds; fu.g_out = bx->g_out; fu.g_T = bx->g_T; fu.g_C = bx->g_C; fu.g_len = bx->g_len;
                    fu.done = bx->done; fu.done_val = bx->done_val;
                    memcpy(fu.qv, bx->q_host, sizeof(float) * (size_t)p.W);              // (W <= 33: scan_fused_supported)
                    if (bx->hint_host) fu.hint_v = *bx->hint_host;
                    fu.done_shards = plan_f.grid < PSH_FUSED_DONE_SHARDS ? plan_f.grid : PSH_FUSED_DONE_SHARDS;
                    bx->taken = true; bx->shards = fu.done_shards;
                }
                if (events) HIP_TRY(hipEventRecord((hipEvent_t)profile->ev_scan_begin, s));
                HIP_TRY(launch_scan_fused(fa, fu, p.aligned, plan_f.grid, s));
                if (events) HIP_TRY(hipEventRecord((hipEvent_t)profile->ev_scan_end, s));
                if (profile) { profile->path = 2; profile->n_sample_rows = (int)rows_f; profile->grid_blocks = plan_f.grid; }
                return PSH_OK;
                }
            }
        }
    }
    Timer tm(stages, s);
    rc = tm.init(); if (rc) return rc;
    rc = tm.mark(); if (rc) return rc;                                       // 0
    PrepArgs pa{queries, qnorm, B, p.qlen, w.qstate, w.total, out_status};  // runs inside the threshold kernel
    rc = tm.mark(); if (rc) return rc;                                       // 1

    Plan plan_s;
    rc = plan_scan(device, p, n_sample > 0 ? n_sample : 1, &plan_s); if (rc) return rc;
    ScanArgs sa = make_scan_args(dataset, queries, p, w, plan_s, row0, stride, n_sample > 0 ? n_sample : 1);
    sa.boot_per_wave = bp.per_wave;
    sa.emb_mx = boot_emx ? 1 : 0;
    sa.blockmax = (use_mx || use_mq) ? w.blockmax : nullptr;
    int n_blockmax = plan_s.grid;
    // the batch's tables and constants (the bootstrap's f16 copies, nx~, the f16 scale; the 8-bit test's step and residue
    // norms): whichever bootstrap follows, the threshold kernel reads the meta words
    if (use_mq) HIP_TRY(launch_mq_prep(queries, B, p.W, w.mq_frag, s));
    // (a single query is better served by the exact bootstrap: 8192 segments are a latency-bound launch either
    // way -- 20.8 vs 18.7 us measured -- and the exact minima admit 9 % fewer candidates)
    if (hint) {
        // the caller's levels: nothing is sampled (the threshold kernel takes tau = tau2 = hint[b])
        n_blockmax = 0;
    } else if (use_mq && bp.per_wave == 1 && boot_mq_supported(p.W)) {
        // segment minima as matrix-core upper bounds (boot_mq_kernel) instead of exact chains
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        const int chunks = boot_mq_chunks(B);
        int64_t gx = (n_sample * sa.nseg + 7) / 8;
        if (gx > ncu) gx = ncu;
        while (gx * chunks > PSH_MAX_BLOCKS) gx /= 2;
        if (gx < 1) gx = 1;
        n_blockmax = (int)gx * chunks;
        sa.mq_frag = w.mq_frag;
        {   // (the rank the threshold kernel will select, below: an estimate's rank -> the minima may be estimates too)
            const int64_t r2e = (4 * (int64_t)k * n_sample + 2 * p.R - 1) / (2 * p.R) + 8;
            sa.boot_estimate = (r2e < k && r2e <= bp.entries) ? 1 : 0;
        }
        HIP_TRY(launch_boot_mq(sa, p.aligned, (int)gx, s));
    } else if (use_lq && bp.per_wave == 1) {
        // upper bounds of the segment minima of every query from the batched long-window kernel (one pass over the sampled rows)
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        sa.q_per_group = scan_lq_chunk(p.W, B);
        sa.n_qgroups = (B + sa.q_per_group - 1) / sa.q_per_group;
        int64_t gx = (n_sample * sa.nseg + 7) / 8;
        if (gx > ncu) gx = ncu;
        if (gx < 1) gx = 1;
        HIP_TRY(launch_scan_lq(sa, PSH_MODE_BOOT, (int)gx, s));
    } else if (rows_path) {
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        int64_t gb = (n_sample + 127) / 128;
        if (gb > 8 * (int64_t)ncu) gb = 8 * (int64_t)ncu;
        sa.boot_wave_min = rows_wave_min ? 1 : 0;
        HIP_TRY(launch_rows(sa, PSH_MODE_BOOT, (int)(gb < 1 ? 1 : gb), s));
    } else {
        HIP_TRY(launch_scan(sa, PSH_MODE_BOOT, p.aligned, plan_s.grid, s));
    }
    rc = tm.mark(); if (rc) return rc;                                       // 2

    // single query on the matrix cores: candidates are filed in two classes around an ESTIMATE of the k-th
    // smallest acc (the rank2-th smallest sampled minimum ~ 2k windows of the whole ensemble below it)
    int rank2 = 0;
    int k_thr = k;             // the rank the threshold kernel selects exactly
    if (hint) {
        // one class of candidates below the caller's level
    } else if (use_mx && mx_estimate) {
        // the number of sampled minima below the ensemble's k-th smallest value is ~ Binomial(entries, k / windows): mean m = k x
        // the sampled fraction, deviation sqrt(m) -- the rank m + 4.5 sqrt(m) + 8 falls short of k windows once in ~10^5 calls
        // (-> status -> the exhaustive pass) and admits ~1.3 k candidates at k = 8192 instead of the 1.6 k of a flat 1.5 m
        const double m = (double)k * (double)n_sample / (double)p.R;
        const int64_t r2 = (int64_t)(m + 4.5 * sqrt(m) + 8.0) + 1;
        if (r2 < k && r2 <= bp.entries) k_thr = (int)r2;
    } else if (use_mx) {
        const int64_t r2 = (2 * (int64_t)k * n_sample + p.R - 1) / p.R + 8;
        rank2 = (r2 < k && r2 <= bp.entries) ? (int)r2 : 0;
    } else {            // every other scan: the embedded ones, one-window rows, the batched matrix-core scan, the VALU-filter scans
        // The embedded scan (and the scan of one-window rows) ADMITS below an estimate: a candidate costs it an exact
        // d x K chain, and the provable tau of a 1/16 sample lets ~16 k of them through.  The estimate is the r2-th
        // smallest sampled minimum, r2 ~ 1.5 k x the sampled fraction (3 k for the thin 1/64 sample of one-window
        // rows) + 16: ~1.5 k (3 k) windows of the whole ensemble lie below it, k only after a 4..14 sigma shortfall.
        // Everything with acc below it is found, so when at least k windows are, the result is the exact top-k; when
        // fewer are, the selection raises the query's status and the caller takes the exhaustive path, as for an
        // overflow.  The threshold kernel selects THIS rank exactly (tau = tau2 = the estimate): read off a bucket edge
        // of the rank-k selection it came out 1.5x too generous, and 25 k candidates per query miss the selection's
        // LDS-resident path (16384 keys) that 12 k take.
        // (the batched matrix-core scan too, at ~2 k: with the provable tau nearly every second group of 4 queries met a
        //  candidate in a segment and left the fast path of its loop -- the handling of survivors was half its instructions)
        // (the thin 1/64 sample of the batched / VALU-filter scans: + 8 on a rank of ~32 -- k windows of the ensemble
        //  put 16 expected minima of the sample below their level, P(Poisson(16) >= 40) = 3e-7 per query; the minima are
        //  upper bounds, which only adds to the margin)
        const bool thin = !p.ker && !rows_path;
        const int64_t r2 = ((p.ker ? 3 : (rows_path ? 6 : 4)) * (int64_t)k * n_sample + 2 * p.R - 1) / (2 * p.R) + (thin ? 8 : 16);
        if (r2 < k && r2 <= bp.entries) k_thr = (int)r2;
    }
    ThresholdArgs ta{w.minbuf, w.min_stride, hint ? 0 : (int)bp.entries, w.qstate, k_thr, 0,
                     ((use_mx || use_mq) && !hint) ? w.blockmax : nullptr, n_blockmax, use_mq ? w.mq_frag : nullptr, mq_i8 ? 1 : 0, rank2, pa, hint};
    HIP_TRY(launch_threshold(ta, B, s));
    rc = tm.mark(); if (rc) return rc;                                       // 3

    Plan plan_f;
    rc = plan_scan(device, p, p.R, &plan_f); if (rc) return rc;
    // a CU-masked stream (psh_stream_create_reserving): one resident round of blocks on the compute units it may use
    if ((flags_of(profile) & PSH_FLAG_RESERVE_CUS) && (flags_of(profile) & PSH_FLAG_OVERLAP) && plan_f.grid > 4 * PSH_STREAM_RESERVED_CUS && !p.ker)
        plan_f.grid -= PSH_STREAM_RESERVED_CUS;
    ScanArgs fa = make_scan_args(dataset, queries, p, w, plan_f, 0, 1, p.R);
    fa.use_mx = use_mx ? 1 : 0;
    fa.bcount2 = (use_mx && rank2 > 0) ? w.bcount2 : nullptr;
    if (use_mx) {
        // scan_mx_kernel reads the fp32 tile only window by window (no sliding refills past the segment):
        // SEG + W - 1 values rounded up to whole float4 stores, padded layout -- every byte counts, the
        // kernel uses 155+ of the 160 KB of LDS
        const int logical = PSH_SEG + p.W + 3;
        fa.tile_floats = (logical + ((logical >> 6) << 2) + 4 + 3) & ~3;
    }
    fa.dbg_times = tuning().dbg_times;
    if (events) HIP_TRY(hipEventRecord((hipEvent_t)profile->ev_scan_begin, s));
    int nblk = plan_f.grid;
    if (use_mq) {
        // one block of 8 waves per CU and query chunk (grid.y); a query's candidates come from the blocks of its
        // chunk only, so slices are indexed by blockIdx.x alone.  (A 16x16x32 layout -- 2 queries x 8 shifts per
        // MFMA, 12 waves per block at 168 VGPRs -- was built and measured: 6.4 ms against 4.7 for the scan.)
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        const int chunks = scan_mq_chunks(B);
        const int64_t n_rs = p.R * ((p.Tp + PSH_SEG - 1) / PSH_SEG);
        int64_t gx = (n_rs + 7) / 8;
        if (gx > ncu) gx = ncu;
        if (gx > PSH_MAX_BLOCKS) gx = PSH_MAX_BLOCKS;
        if (gx < 1) gx = 1;
        (void)chunks;
        nblk = (int)gx;
        fa.mq_frag = w.mq_frag;
        fa.mq_i8 = mq_i8 ? 1 : 0;
        fa.slice = w.cap / nblk;
        HIP_TRY(launch_scan_mq(fa, p.aligned, (int)gx, s));
    } else if (use_lq) {
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        fa.q_per_group = scan_lq_chunk(p.W, B);
        fa.n_qgroups = (B + fa.q_per_group - 1) / fa.q_per_group;
        const int64_t n_rs = p.R * ((p.Tp + PSH_SEG - 1) / PSH_SEG);
        int64_t gx = (n_rs + 7) / 8;
        if (gx > ncu) gx = ncu;
        if (gx > PSH_MAX_BLOCKS) gx = PSH_MAX_BLOCKS;
        if (gx < 1) gx = 1;
        nblk = (int)gx;
        fa.slice = w.cap / nblk;
        HIP_TRY(launch_scan_lq(fa, PSH_MODE_FILTER, nblk, s));
    } else if (rows_path) {
        int ncu = 0;
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        int64_t gb = (p.R + 127) / 128;
        if (gb > 8 * (int64_t)ncu) gb = 8 * (int64_t)ncu;
        if (gb > PSH_MAX_BLOCKS) gb = PSH_MAX_BLOCKS;
        nblk = (int)(gb < 1 ? 1 : gb);
        fa.slice = w.cap / nblk;
        HIP_TRY(launch_rows(fa, PSH_MODE_FILTER, nblk, s));
    } else {
        HIP_TRY(launch_scan(fa, PSH_MODE_FILTER, p.aligned, plan_f.grid, s));
    }
    if (events) HIP_TRY(hipEventRecord((hipEvent_t)profile->ev_scan_end, s));
    rc = tm.mark(); if (rc) return rc;                                       // 4

    SelectArgs se = make_select_args(p, w, out_d, out_idx, out_status, true, nblk, 0);
    se.unsorted_ok = (flags_of(profile) & PSH_FLAG_UNSORTED) ? 1 : 0;
    se.bcount2 = (use_mx && rank2 > 0) ? w.bcount2 : nullptr;
    if (use_mx && rank2 > 0) { se.dataset = dataset; se.queries = queries; se.T = p.T; se.r_offset = p.r_offset; se.W = p.W; }
    se.dbg_times = tuning().dbg_select;
    // (the flags sit behind the B <= 8 totals in their 256-byte slot of the workspace: ints 32 .. 39)
    if (B <= PSH_RANK_MAX_B && !(flags_of(profile) & PSH_FLAG_SELECT_ONE_BLOCK)) {
        int tb = 0;
        while ((1ll << tb) < p.Tp) ++tb;
        se.rank_tbits = ((p.R + p.r_offset) <= (1ll << (32 - tb))) ? tb : -1;   // rows r_offset .. r_offset + R - 1, t < Tp
        se.handled = w.total + 32;
    }
    HIP_TRY(launch_select(se, B, s));
    rc = tm.mark(); if (rc) return rc;                                       // 5

    if (profile) {
        profile->path = 0;
        profile->n_sample_rows = hint ? 0 : (int)n_sample;
        profile->grid_blocks = nblk;           // the blocks whose slices the selection read (psh_candidates_layout)
    }
    if (stages) {
        HIP_TRY(hipStreamSynchronize(s));
        tm.elapsed(0, 1, &profile->prep_ms);
        tm.elapsed(1, 2, &profile->sample_ms);
        tm.elapsed(2, 3, &profile->threshold_ms);
        tm.elapsed(3, 4, &profile->scan_ms);
        tm.elapsed(4, 5, &profile->select_ms);
        tm.elapsed(0, 5, &profile->total_ms);
        rc = max_total(w, B, &profile->n_candidates); if (rc) return rc;
    }
    return PSH_OK;
}

int psh_scan_topk(int device, void* stream, const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                  const float* queries, const float* qnorm, int B, int W, int h, int k,
                  float* out_d, int32_t* out_idx, int32_t* out_status,
                  void* workspace, size_t workspace_bytes, psh_profile* profile) {
    return scan_topk_impl(device, stream, dataset, R, T, r_offset, queries, qnorm, B, W, h, k, nullptr, 0,
                          out_d, out_idx, out_status, workspace, workspace_bytes, profile);
}

int psh_scan_topk_embedded(int device, void* stream, const float* dataset, int64_t R, int64_t T, int64_t r_offset,
                           const float* kernel, int d, int K, const float* hx, const float* hxnorm, int B, int h, int k,
                           float* out_d, int32_t* out_idx, int32_t* out_status,
                           void* workspace, size_t workspace_bytes, psh_profile* profile) {
    if (!kernel || d <= 0) return PSH_ERR_ARG;
    if (T == (int64_t)K + h) return PSH_ERR_UNSUPPORTED;     // one-window rows: psh_embed_rows + psh_scan_topk (see psh.h)
    return scan_topk_impl(device, stream, dataset, R, T, r_offset, hx, hxnorm, B, K, h, k, kernel, d,
                          out_d, out_idx, out_status, workspace, workspace_bytes, profile);
}

int psh_merge_workspace_bytes(int B, int k, size_t* out_bytes) {
    if (!out_bytes || B <= 0 || k <= 0) return PSH_ERR_ARG;
    if (k > PSH_MAX_K) return PSH_ERR_UNSUPPORTED;
    *out_bytes = align_up(sizeof(int2) * (size_t)B * next_pow2(k), 256);
    return PSH_OK;
}

int psh_merge_topk(int device, void* stream, const float* d_lists, const int32_t* idx_lists, int B, int n_in, int k,
                   float* out_d, int32_t* out_idx, void* workspace, size_t workspace_bytes) {
    if (!d_lists || !idx_lists || !out_d || !out_idx || !workspace || B <= 0 || n_in <= 0 || k <= 0) return PSH_ERR_ARG;
    if (k > PSH_MAX_K) return PSH_ERR_UNSUPPORTED;
    if (((uintptr_t)idx_lists & 7u) != 0 || ((uintptr_t)workspace & 7u) != 0) return PSH_ERR_ARG;
    const int kpad = next_pow2(k);
    if (workspace_bytes < sizeof(int2) * (size_t)B * kpad) return PSH_ERR_WORKSPACE;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    SelectArgs s;
    memset(&s, 0, sizeof(s));
    s.cand_d = d_lists;
    s.cand_rt = (const int2*)idx_lists;
    s.cand_stride = n_in;
    s.bcount = nullptr;
    s.n_fixed = n_in;
    s.cap = n_in;
    s.k = k;
    s.kpad = kpad;
    s.skip_negative_rows = 1;
    s.out_d = out_d;
    s.out_idx = out_idx;
    s.sel_rt = (int2*)workspace;
    s.status = nullptr;
    s.qstate = nullptr;
    HIP_TRY(launch_select(s, B, (hipStream_t)stream));
    return PSH_OK;
}

int psh_merge_topk_gathered(int device, void* stream, const float* d_gathered, const int32_t* idx_gathered,
                            int G, int64_t rank_stride, int64_t rank_stride_idx, int B, int k_in, int k,
                            float* out_d, int32_t* out_idx, void* workspace, size_t workspace_bytes) {
    if (!d_gathered || !idx_gathered || !out_d || !out_idx || !workspace || G <= 0 || B <= 0 || k_in <= 0 || k <= 0) return PSH_ERR_ARG;
    if (rank_stride < (int64_t)B * k_in || rank_stride_idx < (int64_t)B * k_in || (int64_t)G * k_in >= (1ll << 31)) return PSH_ERR_ARG;
    if (k > PSH_MAX_K) return PSH_ERR_UNSUPPORTED;
    if (((uintptr_t)idx_gathered & 7u) != 0 || ((uintptr_t)workspace & 7u) != 0) return PSH_ERR_ARG;
    const int kpad = next_pow2(k);
    if (workspace_bytes < sizeof(int2) * (size_t)B * kpad) return PSH_ERR_WORKSPACE;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    SelectArgs s;
    memset(&s, 0, sizeof(s));
    s.cand_d = d_gathered;
    s.cand_rt = (const int2*)idx_gathered;
    s.cand_stride = k_in;              // query b's lists start b * k_in into every rank block
    s.bcount = nullptr;
    s.n_fixed = G * k_in;
    s.list_len = k_in;
    s.list_stride = rank_stride;
    s.list_stride_rt = rank_stride_idx;
    s.cap = G * k_in;
    s.k = k;
    s.kpad = kpad;
    s.skip_negative_rows = 1;
    s.out_d = out_d;
    s.out_idx = out_idx;
    s.sel_rt = (int2*)workspace;
    HIP_TRY(launch_select(s, B, (hipStream_t)stream));
    return PSH_OK;
}

int psh_merge_sorted_gathered(int device, void* stream, const float* d_gathered, const int32_t* idx_gathered,
                              int G, int64_t rank_stride, int64_t rank_stride_idx, int B, int k_in, int k,
                              float* out_d, int32_t* out_idx) {
    if (!d_gathered || !idx_gathered || !out_d || !out_idx || G <= 0 || B <= 0 || k_in <= 0 || k <= 0) return PSH_ERR_ARG;
    if (rank_stride < (int64_t)B * k_in || rank_stride_idx < (int64_t)B * k_in) return PSH_ERR_ARG;
    if (((uintptr_t)idx_gathered & 7u) != 0) return PSH_ERR_ARG;
    if (G > 64 || (int64_t)G * k_in * 4 > 128 * 1024 || (int64_t)k > (int64_t)G * k_in + PSH_MAX_K) return PSH_ERR_UNSUPPORTED;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    MergeSortedArgs m{d_gathered, (const int2*)idx_gathered, rank_stride, rank_stride_idx, G, k_in, k, out_d, out_idx};
    HIP_TRY(launch_merge_sorted(m, B, (hipStream_t)stream));
    return PSH_OK;
}

size_t psh_embed_plan_offset(void) { return PSH_FUSED_BYTES; }

int psh_candidates_layout(int64_t R, int64_t T, int B, int W, int h, int k, size_t workspace_bytes, int64_t* out14) {
    int64_t* out12 = out14;
    if (!out12 || R <= 0 || T <= 0 || B <= 0 || W <= 0 || h < 0 || k <= 0) return PSH_ERR_ARG;
    if (W > PSH_MAX_W || k > PSH_MAX_K) return PSH_ERR_UNSUPPORTED;
    const int64_t Tp = T - W - h + 1;
    if (Tp <= 0) return PSH_ERR_ARG;
    Workspace w;
    char* const base = (char*)(uintptr_t)4096;             // (carve only does pointer arithmetic: any 256-byte aligned base)
    const int rc = carve(base, workspace_bytes, B, k, boot_entries(R, Tp, k), &w);
    if (rc) return rc;
    out12[0] = (char*)w.qstate - base;
    out12[1] = (char*)w.bcount - base;
    out12[2] = (char*)w.bcount2 - base;
    out12[3] = (char*)w.cand_d - base;
    out12[4] = (char*)w.cand_rt - base;
    out12[5] = w.cap;
    out12[6] = (int64_t)offsetof(FusedHdr, cand);
    out12[7] = (int64_t)offsetof(FusedHdr, blk);
    out12[8] = (int64_t)(offsetof(FusedHdr, stream) + offsetof(StreamCtl, ncand));
    out12[9] = PSH_MAX_BLOCKS;
    out12[10] = PSH_FUSED_MAX_BLOCKS;
    out12[11] = PSH_FUSED_FRONT;
    {   // the overlap-friendly launches' lists (w.fused sits at the base)
        void* list = nullptr;
        int per = 0;
        stream_cand_list(w, B, &list, &per);
        out14[12] = (char*)list - base;
        out14[13] = per;
    }
    return PSH_OK;
}

int psh_embedded_supported(int d, int K) {
    if (d <= 0 || K <= 0 || d > PSH_EMB_MAX_D || K > PSH_MAX_W || (int64_t)d * ((K + 3) & ~3) > PSH_EMB_MAX_TAPS) return 0;
    return scan_shmem_bytes(tile_floats_for(K), PSH_MAX_B_PER_LAUNCH, d, K, 512) <= PSH_LDS_BYTES ? 1 : 0;
}

int psh_embed_rows(int device, void* stream, const float* dataset, int64_t R, int64_t T,
                   const float* kernel, int d, int K, float* out) {
    if (!dataset || !kernel || !out || R <= 0 || T <= 0 || d <= 0 || K <= 0 || K > T) return PSH_ERR_ARG;
    if (d > PSH_EMB_MAX_D || (int64_t)d * ((K + 3) & ~3) > PSH_EMB_MAX_TAPS) return PSH_ERR_UNSUPPORTED;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    HIP_TRY(launch_embed_rows(dataset, R, T, kernel, d, K, out, (hipStream_t)stream));
    return PSH_OK;
}

int psh_gather_paths(int device, void* stream, const float* dataset, int64_t R, int64_t C, int64_t T, int64_t r_offset,
                     const int32_t* idx, int64_t n, int len, float* out) {
    if (!dataset || !idx || !out || R <= 0 || C <= 0 || T <= 0 || n < 0 || len <= 0) return PSH_ERR_ARG;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    GatherArgs a{dataset, R, C, T, r_offset, idx, n, (int64_t)len, out};
    HIP_TRY(launch_gather(a, (hipStream_t)stream));
    return PSH_OK;
}

// ---- one BLOCKING shadow() of one Identity query (reference path_shadowing.py:181-218) --------------------------------
int psh_shadow_block_layout(int W, int h, int k, int64_t C, size_t* out7) {
    if (!out7 || W <= 0 || W > PSH_MAX_W || h < 0 || k <= 0 || k > PSH_MAX_K || C <= 0) return PSH_ERR_ARG;
    const size_t n_d = (size_t)4 * k, n_i = (size_t)8 * k, n_p = (size_t)4 * k * (size_t)C * (size_t)(W + h);
    const size_t o_d = PSH_SHADOW_OFF_QUERY + (size_t)4 * PSH_MAX_W;
    const size_t o_i = align_up(o_d + n_d, 256), o_p = align_up(o_i + n_i, 256);
    out7[0] = align_up(o_p + n_p, 256);
    out7[1] = PSH_SHADOW_OFF_STATUS; out7[2] = PSH_SHADOW_OFF_QUERY; out7[3] = PSH_SHADOW_OFF_HINT;
    out7[4] = o_d; out7[5] = o_i; out7[6] = o_p;
    return PSH_OK;
}

static inline double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int psh_shadow_blocking(int device, void* stream, const float* rows, int64_t R, int64_t T, int64_t r_offset,
                        const float* dataset3, int64_t C, int W, int h, int k,
                        void* host_block, size_t host_block_bytes, int with_hint,
                        void* workspace, size_t workspace_bytes, psh_profile* profile) {
    size_t lay[7];
    int rc = psh_shadow_block_layout(W, h, k, C, lay);
    if (rc) return rc;
    if (!host_block || host_block_bytes < lay[0] || !dataset3 || !rows) return PSH_ERR_ARG;
    char* hb = static_cast<char*>(host_block);
    volatile unsigned* done = reinterpret_cast<volatile unsigned*>(hb + PSH_SHADOW_OFF_DONE);
    unsigned* seqp = reinterpret_cast<unsigned*>(hb + PSH_SHADOW_OFF_SEQ);
    const unsigned seq = (*seqp = *seqp + 1u == 0u ? 1u : *seqp + 1u);       // never 0: a fresh block's words
    int32_t* status = reinterpret_cast<int32_t*>(hb + PSH_SHADOW_OFF_STATUS);
    *status = -1;                                                              // (a call that never ran leaves no OK behind)
    psh_profile pf;
    if (profile) pf = *profile; else memset(&pf, 0, sizeof(pf));
    pf.mode = PSH_PROFILE_EVENTS;
    pf.ev_scan_begin = pf.ev_scan_end = nullptr;
    pf.tau_hint = with_hint ? reinterpret_cast<const float*>(hb + PSH_SHADOW_OFF_HINT) : nullptr;
    const double t_begin = now_s();
    BlockingExtras bx{dataset3, reinterpret_cast<float*>(hb + lay[6]), T, (int)C, W + h,
                      const_cast<unsigned*>(done), seq, false, 0,
                      reinterpret_cast<const float*>(hb + PSH_SHADOW_OFF_QUERY), with_hint ? reinterpret_cast<const float*>(hb + PSH_SHADOW_OFF_HINT) : nullptr};
    __atomic_thread_fence(__ATOMIC_SEQ_CST);                                   // query, hint, status: in memory before the doorbell
    rc = scan_topk_impl(device, stream, rows, R, T, r_offset, reinterpret_cast<const float*>(hb + PSH_SHADOW_OFF_QUERY), nullptr,
                        1, W, h, k, nullptr, 0, reinterpret_cast<float*>(hb + lay[4]), reinterpret_cast<int32_t*>(hb + lay[5]), status,
                        workspace, workspace_bytes, &pf, &bx);
    if (profile) { const float* th = profile->tau_hint; const int md = profile->mode; void* e0 = profile->ev_scan_begin; void* e1 = profile->ev_scan_end;
                   *profile = pf; profile->tau_hint = th; profile->mode = md; profile->ev_scan_begin = e0; profile->ev_scan_end = e1; }
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    hipStream_t s = (hipStream_t)stream;
    if (!bx.taken) {
        // not the fused launch (a window or a k it does not serve, a small ensemble): the gather as its own launch, and the
        // stream's end is the call's end
        GatherArgs a{dataset3, R, C, T, r_offset, reinterpret_cast<const int32_t*>(hb + lay[5]), (int64_t)k, (int64_t)(W + h),
                     reinterpret_cast<float*>(hb + lay[6])};
        HIP_TRY(launch_gather(a, s));
        HIP_TRY(hipStreamSynchronize(s));
        return PSH_OK;
    }
    // the launch's completion words (one per shard of its blocks) land in this block as the last thing it writes; a launch
    // that returned early (RETRY before its scan) sets none -- the stream says so
    const double t0 = now_s();
    reinterpret_cast<float*>(hb + PSH_SHADOW_OFF_TIMES)[0] = (float)(1e6 * (t0 - t_begin));     // diagnostics: us spent enqueueing
    double t_query = t0 + 60e-6;
    volatile unsigned* started = reinterpret_cast<volatile unsigned*>(hb + PSH_SHADOW_OFF_STARTED);
    bool seen_start = false, seen_first = false;
    for (unsigned spin = 1;; ++spin) {
        if (!seen_start && *started == seq) { seen_start = true; reinterpret_cast<float*>(hb + PSH_SHADOW_OFF_TIMES)[2] = (float)(1e6 * (now_s() - t0)); }
        bool all = true, any = false;
        for (int i = 0; i < bx.shards; ++i) { const bool d = done[i] == seq; all = all && d; any = any || d; }
        if (any && !seen_first) { seen_first = true; reinterpret_cast<float*>(hb + PSH_SHADOW_OFF_TIMES)[3] = (float)(1e6 * (now_s() - t0)); }
        if (all) break;
        __builtin_ia32_pause();
        if ((spin & 63u) == 0u) {
            const double t = now_s();
            if (t >= t_query) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { snprintf(g_hip_err, sizeof(g_hip_err), "hipStreamQuery -> %s", hipGetErrorString(q)); return PSH_ERR_HIP; }
                t_query = now_s() + 15e-6;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    reinterpret_cast<float*>(hb + PSH_SHADOW_OFF_TIMES)[1] = (float)(1e6 * (now_s() - t0));          // ... and waiting
    return PSH_OK;
}

int psh_count_nonfinite(int device, void* stream, const float* x, int64_t n, unsigned long long* out_count) {
    if (!x || !out_count || n < 0) return PSH_ERR_ARG;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    HIP_TRY(launch_count_nonfinite(x, n, out_count, (hipStream_t)stream));
    return PSH_OK;
}

int psh_rows_nonfinite(int device, void* stream, const float* dataset, int64_t R, int64_t C, int64_t T, int32_t* out_flags) {
    if (!dataset || !out_flags || R < 0 || C <= 0 || T <= 0) return PSH_ERR_ARG;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    HIP_TRY(launch_rows_nonfinite(dataset, R, C * T, out_flags, (hipStream_t)stream));
    return PSH_OK;
}

int psh_smear_nonfinite(int device, void* stream, const float* dataset, int64_t R, int64_t C, int64_t T, int back, int fwd, float* out) {
    if (!dataset || !out || R < 0 || C <= 0 || T <= 0 || back < 0 || fwd < 0) return PSH_ERR_ARG;
    if (R * T >= ((int64_t)1 << 31) * 256) return PSH_ERR_UNSUPPORTED;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    HIP_TRY(launch_smear_nonfinite(dataset, R, C, T, back, fwd, out, (hipStream_t)stream));
    return PSH_OK;
}

int psh_weighted_moments(int device, void* stream, const float* values, const double* weights, int B, int k, int m,
                         double* out_mean, double* out_std) {
    if (!values || !out_mean || !out_std || B <= 0 || k <= 0 || m <= 0) return PSH_ERR_ARG;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    MomentsArgs a{values, weights, B, k, m, out_mean, out_std};
    HIP_TRY(launch_moments(a, (hipStream_t)stream));
    return PSH_OK;
}

int psh_realized_variance(int device, void* stream, const float* x, int64_t n_rows, int64_t row_stride, int len,
                          const int* Ts, int nT, int vol, float* out) {
    if (!x || !Ts || !out || n_rows < 0 || len <= 0 || row_stride < len || nT <= 0) return PSH_ERR_ARG;
    if (nT > PSH_RV_MAX_T) return PSH_ERR_UNSUPPORTED;
    if (n_rows == 0) return PSH_OK;
    DeviceGuard g(device);
    if (!g.ok) { snprintf(g_hip_err, sizeof(g_hip_err), "hipSetDevice(%d) failed", device); return PSH_ERR_HIP; }
    RvArgs a{};
    a.x = x; a.n_rows = n_rows; a.row_stride = row_stride; a.nT = nT; a.vol = vol ? 1 : 0; a.out = out;
    for (int i = 0; i < nT; ++i) {
        if (Ts[i] <= 0) return PSH_ERR_ARG;
        a.Ts[i] = Ts[i] < len ? Ts[i] : len;                 // numpy: x[..., :T] clips to the row
    }
    HIP_TRY(launch_realized_variance(a, (hipStream_t)stream));
    return PSH_OK;
}

}  // extern "C"
