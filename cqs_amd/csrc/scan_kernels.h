// scan_kernels.h — launch interface of the gfx950 scan + top-k kernels.
// Internal to libcqs_hip.so (the public boundary is include/cqs_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cqs {

// Geometry shared by host and device code.
constexpr uint32_t kTaskRows = 64;        // rows one wave scores per work-queue task (large corpora; MFMA path)
constexpr uint32_t kTaskRowsSmall = 16;   // small corpora: finer tasks so every CU gets several waves
constexpr uint32_t kRowsPerBlock = 256;   // n_pad granule (score row stride)
constexpr uint32_t kHistBins = 4096;      // threshold-search histogram bins
constexpr uint32_t kCandCap = 8192;       // candidates one sort block holds in LDS
constexpr uint32_t kMaxK = 1024;
constexpr uint32_t kMaxGemvQ = 8;         // queries per HBM-streaming scan pass
constexpr uint32_t kWorkWords = 64;       // work-queue heads (one per scan launch of a search); zero on entry,
                                          // re-zeroed by select_finish

struct ScanArgs {
    const float* rows;      // [n, dim] f32, row-major, HBM
    uint32_t n;             // rows in this index/shard
    uint32_t n_pad;         // n rounded up to kRowsPerBlock (score row stride)
    uint32_t dim;
    const float* q;         // [b, dim] f32, device
    uint32_t b;             // queries
    float* scores;          // [b, n_pad] f32, device
    const uint32_t* keep;   // nullable device bitset, ceil(n/32) words
    uint32_t mode;          // CQS_HIP_MODE_*
    float threshold;
    bool nontemporal;       // stream the corpus past L2 (corpus >> Infinity Cache)
    bool linear_bins;       // scores bounded in [-1,1] (cosine / pipeline mode): linear histogram bins
    uint32_t k;
    float* gmax;            // [b, n_pad/group_rows] per-group maxima (written by the scan)
    uint32_t group_rows;    // kTaskRows or kTaskRowsSmall: rows per work-queue task = rows per maxima group
    uint32_t* work;         // [kWorkWords] work-queue heads (zero on entry)
    uint32_t n_cu;          // compute units of the device
    void* dbg;              // nullable: 16 x u64 phase stamps of select_finish (CQS_HIP_DEBUG_STAMPS=1)
};

// scores[q][row] = dot(rows[row], q) (+ mode / bitset / non-finite handling; dropped
// entries = -inf), gmax[q][g] = max of 64-row group g.
hipError_t launch_scan(const ScanArgs& a, hipStream_t stream);

// Exact top-k of each query's score row (one workgroup per query; see select_finish_kernel).
// Output: out_keys[b*k] packed (ordered(score)<<32 | ~global_row) sorted descending,
// out_counts[b].  Leaves the work-queue heads zeroed for the next search.
hipError_t launch_select(const ScanArgs& a, uint32_t row_base, uint64_t* out_keys, uint32_t* out_counts,
                         hipStream_t stream);

// Batched path (scan_mfma.hip): query block [q0, q0+nq), nq <= 256, on the f32 matrix cores.
// a.q must be readable and zero-padded up to mfma_query_tile(nq) rows past q0.
hipError_t launch_scan_mfma(const ScanArgs& a, uint32_t q0, uint32_t nq, uint32_t work_slot, hipStream_t stream);
uint32_t mfma_query_tile(uint32_t nq);
constexpr uint32_t kMfmaMinQueries = 9;   // below this the HBM-streaming gemv passes win
inline bool use_mfma(uint32_t b, uint32_t dim) { return b >= kMfmaMinQueries && dim % 32u == 0; }

bool scan_dim_supported(uint32_t dim);

}  // namespace cqs
