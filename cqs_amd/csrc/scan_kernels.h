// scan_kernels.h — launch interface of the gfx950 scan + top-k kernels.
// Internal to libcqs_hip.so (the public boundary is include/cqs_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cqs {

// Geometry shared by host and device code.
constexpr uint32_t kRowsPerBlock = 256;   // scan: 4 waves x 64 rows; n_pad granule
constexpr uint32_t kHistBins = 4096;      // 12-bit radix digits
constexpr uint32_t kCandCap = 8192;       // candidates one sort block holds in LDS
constexpr uint32_t kMaxK = 1024;
constexpr uint32_t kMaxGemvQ = 8;         // queries per HBM-streaming scan pass
// Per-query select scratch, in u32 words: hist1[4096] hist2[4096] cand_count pad[3]
constexpr uint32_t kSelWords = 2 * kHistBins + 4;

struct ScanArgs {
    const float* rows;      // [n, dim] f32, row-major, HBM
    uint32_t n;             // rows in this index/shard
    uint32_t n_pad;         // n rounded up to kRowsPerBlock (score row stride)
    uint32_t dim;
    const float* q;         // [b, dim] f32, device
    uint32_t b;             // queries in this launch group
    float* scores;          // [b, n_pad] f32, device
    const uint32_t* keep;   // nullable device bitset, ceil(n/32) words
    uint32_t mode;          // CQS_HIP_MODE_*
    float threshold;
    bool nontemporal;       // stream the corpus past L2 (corpus >> Infinity Cache)
};

// scores[q][row] = dot(rows[row], q) (+ mode / bitset / non-finite handling,
// dropped entries = -inf).  Returns hipSuccess or the launch error.
hipError_t launch_scan(const ScanArgs& a, hipStream_t stream);

// Exact top-k of each query's score row: radix threshold search (2 fused
// 12-bit levels) -> candidate collect -> one-block bitonic sort, with an
// in-kernel exact fallback for heavy ties.  `sel` is b*kSelWords u32 of
// scratch (zeroed here), `cand` b*kCandCap u64.  Output: out_keys[b*k] packed
// (ordered(score)<<32 | ~global_row) sorted descending, out_counts[b].
hipError_t launch_select(const float* scores, uint32_t n, uint32_t n_pad, uint32_t b, uint32_t k,
                         uint32_t row_base, uint32_t* sel, uint64_t* cand,
                         uint64_t* out_keys, uint32_t* out_counts, hipStream_t stream);

bool scan_dim_supported(uint32_t dim);

}  // namespace cqs
