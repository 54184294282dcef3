// scan_kernels.h — launch interface of the gfx950 scan + top-k kernels.
// Internal to libcqs_hip.so (the public boundary is include/cqs_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cqs {

// Geometry shared by host and device code.
constexpr uint32_t kTaskRows = 64;        // rows of the largest work-queue task (= maxima group of the MFMA path)
constexpr uint32_t kTaskRowsSmall = 16;   // rows of the smallest task (bounds the number of maxima groups)
constexpr uint32_t kRowsPerBlock = 256;   // n_pad granule (score row stride)
constexpr uint32_t kHistBins = 4096;      // threshold-search histogram bins
constexpr uint32_t kCandCap = 8192;       // candidates one sort block holds in LDS
constexpr uint32_t kMaxK = 1024;
constexpr uint32_t kMaxGemvQ = 8;         // queries per HBM-streaming scan pass
constexpr uint32_t kDbgWaves = 4096;      // per-wave start/end stamps kept by the scan when ScanArgs.dbg is set
constexpr uint32_t kWorkWords = 64;       // work-queue heads (one per scan launch of a search); zero on entry,
                                          // re-zeroed by select_finish

// Work-queue tasks in row order: nA tasks of 64 rows, then nB of 32, then nC of 16 (64 nA + 32 nB +
// 16 nC = n_pad).  One task = one wave's unit of work = one group of the select's maxima index.
struct TaskTiers {
    uint32_t nA, nB, nC;
    __host__ __device__ uint32_t total() const { return nA + nB + nC; }
    // first row and row count of task t
    __host__ __device__ uint32_t locate(uint32_t t, uint32_t& rows) const {
        if (t < nA) { rows = 64u; return t * 64u; }
        if (t < nA + nB) { rows = 32u; return nA * 64u + (t - nA) * 32u; }
        rows = 16u;
        return nA * 64u + nB * 32u + (t - nA - nB) * 16u;
    }
};
TaskTiers plan_tiers(uint32_t n_pad, uint32_t n_cu, bool uniform64);

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
    bool range_bins = false;// select only: linear bins over the range of the row's own group maxima (the sparse index)
    uint32_t k;
    float* gmax;            // [b, tiers.total()] per-task maxima (written by the scan)
    uint64_t* gaux = nullptr;// nullable, [b, tiers.total()]: (lane of the task's maximum << 32) | f32 bits of its runner-up
                            // (written by the gemv scan and the sparse index; the matrix-core scan does not: its select
                            // launch ignores the field)
    TaskTiers tiers;        // plan_tiers(n_pad, n_cu, uniform_groups(b, dim))
    uint32_t* work;         // [kWorkWords] work-queue heads (zero on entry)
    uint32_t n_cu;          // compute units of the device
    bool gemv_only = false; // never the matrix-core kernel, whatever b: blocks of > 8 queries run as passes of <= 8
                            // (the scores of a gemv pass do not depend on how many queries share it)
    void* dbg;              // nullable: (16 + 2 * kDbgWaves) x u64: select_finish phase stamps, then the scan's
                            // per-wave start/end stamps (CQS_HIP_DEBUG_STAMPS=1)
};

// scores[q][row] = dot(rows[row], q) (+ mode / bitset / non-finite handling; dropped
// entries = -inf), gmax[q][t] = max over the rows of task t (a.tiers).
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
inline bool uniform_groups(uint32_t b, uint32_t dim) { return use_mfma(b, dim); }  // kernels that need 64-row groups only

bool scan_dim_supported(uint32_t dim);

#if defined(__HIPCC__)
// Maximum over the 64 lanes of a wave, in every lane, without the LDS: four DPP steps inside the 16-lane rows (quad_perm
// xor 1, xor 2, row_half_mirror, row_mirror), then v_permlane16_swap and v_permlane32_swap between the rows.  (`__shfl_xor`
// compiles to ds_bpermute_b32 on gfx950: six dependent trips through the LDS queue per maximum.)  max16: the first four
// steps alone = the maximum of each aligned group of 16 lanes.
template <int CTRL>
__device__ __forceinline__ float wave_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_max16(float m) {
    m = fmaxf(m, wave_dpp<0xB1>(m));
    m = fmaxf(m, wave_dpp<0x4E>(m));
    m = fmaxf(m, wave_dpp<0x141>(m));
    m = fmaxf(m, wave_dpp<0x140>(m));
    return m;
}
__device__ __forceinline__ float wave_max64(float m) {
    typedef unsigned wm_u2 __attribute__((ext_vector_type(2)));
    m = wave_max16(m);
    const wm_u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const wm_u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
#endif

}  // namespace cqs
