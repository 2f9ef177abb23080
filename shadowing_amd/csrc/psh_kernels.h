// psh_kernels.h -- argument blocks and launch entry points shared by the kernels
// (psh_kernels.hip) and the C ABI (psh_capi.hip).  Internal; the public header is
// include/psh.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PSH_L 16                     // consecutive windows per lane
#define PSH_SEG (64 * PSH_L)         // windows per wave-segment
#define PSH_NSTAGE 5                 // 16-byte loads per lane per segment: ceil((SEG + W_max - 1) / 256)
#define PSH_SCAN_THREADS 256
#define PSH_SELECT_THREADS 1024
#define PSH_NBINS 2048               // log-spaced histogram bins per query

#define PSH_MODE_SAMPLE 0
#define PSH_MODE_FILTER 1
#define PSH_MODE_ALL 2

#define PSH_STATUS_OK_ 0
#define PSH_STATUS_OVERFLOW_ 1

namespace psh {

struct QueryState {   // one per query, device
    float xn;         // ||x||
    float tau;        // admission threshold on acc (exclusive)
    int base;         // histogram key base
    int n_valid;      // valid entries in out_d/out_idx after the last select
};

struct PrepArgs {
    const float* queries;
    const float* qnorm_in;   // nullable
    int B, W;
    QueryState* qstate;
    int* counts;
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
    int64_t r_offset;
    const float* queries;    // B x W
    int B;
    int n_qgroups, q_per_group;
    int tile_floats;         // LDS floats per wave
    QueryState* qstate;
    unsigned* hist;          // B x PSH_NBINS          (SAMPLE)
    float* cand_d;           // B x cap                (FILTER / ALL)
    int2* cand_rt;           // B x cap
    int* counts;             // B
    int cap;
};

struct ThresholdArgs {
    const unsigned* hist;
    QueryState* qstate;
    int k;
};

struct SelectArgs {
    const float* cand_d;
    const int2* cand_rt;
    int64_t cand_stride;     // elements between queries
    const int* counts;       // nullable -> n_fixed
    int n_fixed;
    int cap;
    int k, kpad;
    int skip_negative_rows;  // merge: entries with r < 0 are padding
    float* out_d;
    int32_t* out_idx;
    int2* sel_rt;            // B x kpad scratch
    int* status;             // nullable
    QueryState* qstate;      // nullable
};

struct ReseedArgs {
    const float* out_d;
    const int32_t* out_idx;
    QueryState* qstate;
    float* cand_d;
    int2* cand_rt;
    int* counts;
    int cap, k;
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
hipError_t scan_blocks_per_cu(int W, bool aligned, size_t shmem, int* out);
hipError_t launch_threshold(const ThresholdArgs& a, int B, hipStream_t s);
hipError_t launch_select(const SelectArgs& a, int B, hipStream_t s);
hipError_t launch_reseed(const ReseedArgs& a, int B, hipStream_t s);
hipError_t launch_gather(const GatherArgs& a, hipStream_t s);

}  // namespace psh
