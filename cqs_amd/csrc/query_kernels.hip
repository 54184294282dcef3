// query_kernels.hip — the search-time forward: ONE short sequence (<= 64 tokens) through EmbeddingGemma.
//
// The reference embeds the query text on every search (`Embedder::embed_query`, src/embedder/core.rs:768-856, called at
// src/cli/commands/search/query.rs:595; README.md:1079-1082 quotes ~3 ms on its CUDA EP).  At one sequence of 8-32 tokens
// the forward is not a throughput problem: 2 x 101.5 M parameters are 212 MB of bf16 weights streamed once (~30 us at
// HBM / Infinity-Cache rates) and ~10 MFLOP per token of matrix work.  The batch chain (embedder.hip: run_layers) serves
// that shape with 9-10 launches per layer, each a tile kernel built for 16 384 tokens: 230 dependent launches of 4-8 us,
// 1.3 ms.  What bounds a kernel here is LATENCY: a launch boundary (~1.3 us, MI355X_MICROARCH.md price list
// "boundary"), one round trip to L2 / MALL for the activations the previous kernel wrote, one for the weights.  So:
//
//   * 4 launches per layer, the minimum the data flow allows with every projection's N range spread over the chip (a
//     row norm needs the whole row = all N-slices of the producing GEMM, so every GEMM output is a grid-wide seam; a
//     grid barrier inside one launch costs 4-5 us on this chip, 3x a launch boundary - not a persistent kernel):
//         QKV      [ x += norm(down_prev)(1+w) ; xn = norm(x)(1+w) ]  -> qkv = xn Wqkv^T
//         attention[ k norm + rope, q norm + rope + scale ] + o_proj  -> y = softmax(q k^T) v Wo^T   (every o_proj workgroup
//                                                                        redoes the attention of all heads in its own LDS)
//         GeGLU    [ x += norm(y)(1+w) ; xn = norm(x)(1+w) ]          -> h = gelu(xn Wg^T) * (xn Wu^T)
//         down                                                          -> y = h Wd^T
//     the row-wise add + RMSNorm pairs live in the PROLOGUE of the GEMM that consumes them: every workgroup recomputes
//     them for its rows instead of a launch + a round trip each; the workgroup of column tile 0 also writes the new residual
//     stream to the other of two x buffers (everybody reads the old one: no race);
//   * a GEMM workgroup = 8 or 16 output columns x a block of rows, its 4 or 8 waves split K (partial tiles summed through
//     LDS); weight slice and activation rows arrive by COALESCED loads and go through LDS (a fragment gather of 16 rows x
//     64 B costs ~44 clocks of address processing per wave-instruction against ~16 for a contiguous 1 KB);
//     v_mfma_f32_16x16x32_bf16 with the weight rows as the A operand (a lane ends with 4 consecutive output columns of
//     one token row);
//   * what a kernel costs is (a) its executed instruction count (one wave per SIMD: template parameters instead of guards,
//     DPP + readlane reductions) and (b) the bytes ONE workgroup pulls through its CU's L2 port (~70 GB/s for lines every
//     workgroup shares): from 5 tokens on the rows are cut into blocks of 8 (<= 16 tokens) or 16 rows (blockIdx.y) x
//     16-column tiles, so a workgroup normalises only its block;
//   * the head (final add + norm, mean pool, Dense 768 -> 3072 -> 768) is two more launches of the same kernels;
//   * T is a launch parameter (no row past the query's length is loaded; the engine keeps one captured hipGraph per length).
// K order differs from the batch kernels (K split over a workgroup's waves) -> results agree with the batch path to bf16
// rounding noise (cosine >= 0.9999, tests/test_query_path_gpu.py), not bit for bit.  Numbers: DESIGN.md 3.8.
#include "embed_kernels.h"
#include "launch_util.h"

#include <cmath>
#include <cstdlib>

namespace cqs {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned qf_u2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kQfPad = 8;            // bf16 elements of row padding in the LDS activation tile

template <int CTRL>
__device__ __forceinline__ float qf_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the wave: four DPP steps inside each 16-lane row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror),
// then the four row sums through v_readlane (uniform result; no LDS round trip as with ds_bpermute shuffles)
__device__ __forceinline__ float qf_wave_sum(float v) {
    v += qf_dpp<0xB1>(v);
    v += qf_dpp<0x4E>(v);
    v += qf_dpp<0x141>(v);
    v += qf_dpp<0x140>(v);
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}
__device__ __forceinline__ float qf_xor32(float v, int lane) {       // the value of lane ^ 32
    const qf_u2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(lane < 32 ? a[1] : a[0]);
}
__device__ __forceinline__ float qf_xor16_max(float v) {
    const qf_u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float qf_xor32_max(float v) {
    const qf_u2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float qf_xor16_sum(float v) {
    const qf_u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float qf_xor32_sum(float v) {
    const qf_u2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float qf_gelu_tanh(float x) {      // as gelu_tanh (embed_kernels.hip)
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = 2.0f * k0 * (x + k1 * x * x * x);
    return x * __frcp_rn(1.0f + __expf(-u2));
}
__device__ __forceinline__ float qf_gelu_erf(float x) {       // as gelu_erf (embed_kernels.hip): Abramowitz-Stegun 7.1.26
    const float z = __builtin_fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + __builtin_copysignf(e, x));
}
// bias + activation of the plain / staged kernels' epilogue (columns col .. col + 3)
__device__ __forceinline__ f4 qf_bias_act(f4 v, const float* __restrict__ bias, int32_t act, uint32_t col) {
    if (bias) v += *(const f4*)(bias + col);
    if (act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = qf_gelu_erf(v[r]);
    }
    return v;
}

// Diagnostic stamps (off unless the engine was created under CQS_HIP_QUERY_STAMPS=1): workgroup b (< 256) of chain
// kernel `slot` writes the 100 MHz realtime counter at phase i - where a 4 us kernel that moves 30 KB spends its time.
#define QF_STAMP(P, i)                                                                                      \
    do {                                                                                                    \
        if ((P).dbg && threadIdx.x == 0 && blockIdx.x < 256u)                                               \
            (P).dbg[((size_t)(P).dbg_slot * 256u + blockIdx.x) * 8u + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

enum { QF_PRO_NONE = 0, QF_PRO_EMBED = 1, QF_PRO_ADDNORM = 2, QF_PRO_POOL = 3,
       QF_PRO_ADDLN = 4 /* BERT: x = LayerNorm(xb_in + y) gamma + beta (bf16 stream), the GEMM's activations = x */ };
enum { QF_EPI_BF16 = 0, QF_EPI_GEGLU = 1, QF_EPI_F32 = 2 };

struct QfGemmParams {
    const int32_t* tok;       // EMBED: token ids [T]
    const bf16_t* emb;        // EMBED: token table [vocab, H]
    float scale;              // EMBED: sqrt(H)
    const float* x_in;        // ADDNORM / POOL: residual stream [64, H] f32
    const bf16_t* y;          // ADDNORM / POOL: branch output [64, H] bf16
    const float* w_post;      // ADDNORM / POOL: post-branch norm weight [H]
    const float* w_next;      // next pre-norm weight [H] (POOL: the model's final norm)
    float* x_out;             // EMBED / ADDNORM: new residual stream [64, H] (written by workgroup 0; != x_in)
    float eps;
    const bf16_t* A;          // PRO_NONE: activations [64, K] bf16 (one_row: [K])
    const bf16_t* W;          // [N, K] bf16
    void* C;                  // [64, ldc] bf16 / f32 (POOL and its successor: one row)
    uint32_t K, ldc;
    uint32_t T;               // tokens, 1..64: a LAUNCH parameter (the engine keeps one captured graph per length)
    int32_t one_row;          // 1: the GEMM's activations are ONE row (the pooled vector)
    const float* bias;        // nullable, [N] f32 added before the activation (BERT projections; not with GEGLU)
    int32_t act;              // 1 = erf-GELU after the bias (BERT's FFN)
    const bf16_t* xb_in;      // ADDLN: the bf16 residual stream [rows, H]; w_post = gamma, w_next = beta, eps
    bf16_t* xb_out;           // ADDLN: the new stream (written by the workgroups of column tile 0; != xb_in)
    unsigned long long* dbg;  // nullable (CQS_HIP_QUERY_STAMPS=1): [kernel slot][workgroup < 256][8] realtime stamps
    uint32_t dbg_slot;
};

// What these kernels are built around (measured with the stamps below, then read off the ISA): at one wave per SIMD a
// wave issues one instruction per ~4-5 clocks and every launch runs its code exactly once, so a kernel's time is its
// EXECUTED INSTRUCTION COUNT.  The first versions (runtime row / tile counts behind guards inside unrolled loops) were
// 10-16 KB of straight-line code, 2 000 instructions of which ~1 100 were register copies at the guards' joins: 3-5 us
// per kernel whatever the bytes.  Hence: row-tile count MT, rows per wave RB and k-steps per batch CH are TEMPLATE
// parameters (no guards, no copies), dead slots re-do a real row instead of being predicated (same address: no extra
// bytes; same value stored to the same place), weight rows past NC repeat a real row (their output columns are never
// stored).

// ---- C = A W^T for activations that already exist as bf16 rows (o_proj, down, Dense 2) -------------------------------
// One workgroup = NC output columns x all rows; its 4 waves split K; a wave walks its K range in `nb` batches of CH
// k-steps, every load of a batch in flight before the first MFMA (K <= 1152: one batch).
template <int MT, int CH, int EPI, int NC>
__global__ __launch_bounds__(256) void qf_gemm_plain_kernel(const QfGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qf_smem[];
    float* const red = (float*)qf_smem;                            // [4 waves][MT][64 lanes] f4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    QF_STAMP(p, 0);
    const uint32_t rows = p.one_row ? 1u : p.T;
    const uint32_t K = p.K, kw = K / 4u, nb = kw / (32u * (uint32_t)CH);
    const uint32_t koff = (uint32_t)wid * kw + 8u * (uint32_t)lg;
    const bf16_t* wp = p.W + (size_t)(blockIdx.x * (uint32_t)NC + (uint32_t)(l15 % NC)) * K + koff;
    const bf16_t* ap[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const uint32_t r = 16u * (uint32_t)m + (uint32_t)l15;
        ap[m] = p.A + (size_t)(r < rows ? r : rows - 1u) * K + koff;
    }
    f4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f4)(0.f);
    for (uint32_t b = 0; b < nb; ++b) {
        bf8 wf[CH], af[MT][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            wf[u] = *(const bf8*)(wp + 32 * u);
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m][u] = *(const bf8*)(ap[m] + 32 * u);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[m][u], acc[m], 0, 0, 0);
        wp += 32 * CH;
#pragma unroll
        for (int m = 0; m < MT; ++m) ap[m] += 32 * CH;
    }
    QF_STAMP(p, 1);
    // sum the four K-quarters through LDS; wave w finishes m-tile w.  acc[m][r] = C[row 16 m + l15][col 4 lg + r]
#pragma unroll
    for (int m = 0; m < MT; ++m) *(f4*)(red + ((size_t)(wid * MT + m) * 64 + lane) * 4) = acc[m];
    __syncthreads();
    QF_STAMP(p, 2);
    if (wid >= MT) return;
    f4 v = *(const f4*)(red + ((size_t)wid * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *(const f4*)(red + ((size_t)(w * MT + wid) * 64 + lane) * 4);
    const uint32_t row = 16u * (uint32_t)wid + (uint32_t)l15;
    if (row >= rows || 4 * lg >= NC) return;                       // columns 4 lg .. 4 lg + 3 of the tile: real iff < NC
    const size_t off = (size_t)row * p.ldc + blockIdx.x * (uint32_t)NC + 4u * (uint32_t)lg;
    v = qf_bias_act(v, p.bias, p.act, blockIdx.x * (uint32_t)NC + 4u * (uint32_t)lg);
    if (EPI == QF_EPI_F32) {
        *(f4*)((float*)p.C + off) = v;
    } else {
        bf4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
        *(bf4*)((bf16_t*)p.C + off) = o;
    }
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); QF_STAMP(p, 3); }
}

// ---- the same with both operands staged through LDS by COALESCED loads -----------------------------------------------------
// The kernel above feeds the MFMA straight from global memory: every fragment load is a gather of 16 rows x 64 B, ~44
// clocks of address processing per wave-instruction against ~16 for a contiguous 1 KB (measured with the stamps: with
// the same number of loads made contiguous - wrong data, timing only - `down` went from 2.05 to 1.05 us at 8 tokens, 3.3 to
// 2.0 at 32).  Here thread (row group rg = tid / 32, l32 = tid % 32) copies 16-byte chunks l32, l32 + 32, ... of rows rg,
// rg + 8, ... of the weight slice (NC consecutive rows = one contiguous block) and of the activation rows: a wave-instruction
// covers 2 rows x 512 contiguous bytes.  LDS rows are padded by 16 B (K % 128 == 0: consecutive rows land 4 banks apart).
// Chunks / rows past the end re-copy the last real one (same data to the same place: no guards in the unrolled code).
// KC32 = ceil(K / 8 / 32) chunk iterations per row, CH k-steps per batch (CH x batches = K / 4 / 32), ONE: the activations
// are one row (Dense 2).
template <int MT, int CH, int KC32, int EPI, int NC, int ONE>
__global__ __launch_bounds__(256) void qf_gemm_staged_kernel(const QfGemmParams p) {
    static_assert(NC == 8, "one 8-row weight slice per workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char qf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    QF_STAMP(p, 0);
    // blockIdx.y = block of 16 MT rows (33-64 tokens run as two blocks of 32: 64 rows of K = 1152 do not fit LDS beside the
    // weight slice, and the gather kernel they used to take pays ~44 clocks of address processing per fragment load)
    const uint32_t row0 = ONE ? 0u : blockIdx.y * (uint32_t)(16 * MT);
    const uint32_t rows = ONE ? 1u : (p.T - row0 < (uint32_t)(16 * MT) ? p.T - row0 : (uint32_t)(16 * MT));
    const uint32_t K = p.K, kc = K / 8u, kw = K / 4u, nb = kw / (32u * (uint32_t)CH);
    const uint32_t ldk = K + 8u;                                   // LDS row stride (elements)
    bf16_t* const sW = (bf16_t*)qf_smem;                           // [NC][ldk]
    bf16_t* const sA = sW + (size_t)NC * ldk;                      // [16 MT][ldk]  (ONE: [1][ldk])
    constexpr int AR = ONE ? 1 : 16 * MT;
    float* const red = (float*)(sA + (size_t)AR * ldk);            // [4 waves][MT][64 lanes] f4

    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint32_t rg = (uint32_t)tid >> 5, l32 = (uint32_t)tid & 31u;
    constexpr int AI = ONE ? 0 : 2 * MT;                           // activation row iterations of 8 rows
    u4 wreg[KC32], areg[AI > 0 ? AI : 1][KC32], oreg[ONE ? (KC32 + 7) / 8 : 1];
    const bf16_t* wsrc = p.W + (size_t)(blockIdx.x * (uint32_t)NC + rg) * K;
#pragma unroll
    for (int j = 0; j < KC32; ++j) {
        const uint32_t c = l32 + 32u * (uint32_t)j < kc ? l32 + 32u * (uint32_t)j : kc - 1u;
        wreg[j] = *(const u4*)(wsrc + c * 8u);
    }
    if (ONE) {                                                      // the one activation row: chunk tid, tid + 256, ...
#pragma unroll
        for (int j = 0; j < (KC32 + 7) / 8; ++j) {
            const uint32_t c = (uint32_t)tid + 256u * (uint32_t)j < kc ? (uint32_t)tid + 256u * (uint32_t)j : kc - 1u;
            oreg[j] = *(const u4*)(p.A + c * 8u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const uint32_t r = 8u * (uint32_t)i + rg;
            const bf16_t* asrc = p.A + (size_t)(row0 + (r < rows ? r : rows - 1u)) * K;
#pragma unroll
            for (int j = 0; j < KC32; ++j) {
                const uint32_t c = l32 + 32u * (uint32_t)j < kc ? l32 + 32u * (uint32_t)j : kc - 1u;
                areg[i][j] = *(const u4*)(asrc + c * 8u);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < KC32; ++j) {
        const uint32_t c = l32 + 32u * (uint32_t)j < kc ? l32 + 32u * (uint32_t)j : kc - 1u;
        *(u4*)(sW + (size_t)rg * ldk + c * 8u) = wreg[j];
    }
    if (ONE) {
#pragma unroll
        for (int j = 0; j < (KC32 + 7) / 8; ++j) {
            const uint32_t c = (uint32_t)tid + 256u * (uint32_t)j < kc ? (uint32_t)tid + 256u * (uint32_t)j : kc - 1u;
            *(u4*)(sA + c * 8u) = oreg[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const uint32_t r = 8u * (uint32_t)i + rg;
            const uint32_t rr = r < rows ? r : rows - 1u;
#pragma unroll
            for (int j = 0; j < KC32; ++j) {
                const uint32_t c = l32 + 32u * (uint32_t)j < kc ? l32 + 32u * (uint32_t)j : kc - 1u;
                *(u4*)(sA + (size_t)rr * ldk + c * 8u) = areg[i][j];
            }
        }
    }
    __syncthreads();
    QF_STAMP(p, 1);
    // fragments: weight row l15 % NC (rows past NC repeat real rows: their output columns are never stored), activation
    // row 16 m + l15 (rows past `rows` hold whatever LDS held: they only feed output columns that are never stored)
    const uint32_t koff = (uint32_t)wid * kw + 8u * (uint32_t)lg;
    const bf16_t* wp = sW + (size_t)(l15 % NC) * ldk + koff;
    const bf16_t* ap = sA + (size_t)(ONE ? 0 : l15) * ldk + koff;
    f4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f4)(0.f);
    for (uint32_t b = 0; b < nb; ++b) {
        bf8 wf[CH], af[MT][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            wf[u] = *(const bf8*)(wp + 32 * u);
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m][u] = *(const bf8*)(ap + (size_t)(ONE ? 0 : 16 * m) * ldk + 32 * u);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[m][u], acc[m], 0, 0, 0);
        wp += 32 * CH;
        ap += 32 * CH;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) *(f4*)(red + ((size_t)(wid * MT + m) * 64 + lane) * 4) = acc[m];
    __syncthreads();
    QF_STAMP(p, 2);
    if (wid >= MT) return;
    f4 v = *(const f4*)(red + ((size_t)wid * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *(const f4*)(red + ((size_t)(w * MT + wid) * 64 + lane) * 4);
    const uint32_t row = 16u * (uint32_t)wid + (uint32_t)l15;
    if (row >= rows || 4 * lg >= NC) return;
    const size_t off = (size_t)(row0 + row) * p.ldc + blockIdx.x * (uint32_t)NC + 4u * (uint32_t)lg;
    v = qf_bias_act(v, p.bias, p.act, blockIdx.x * (uint32_t)NC + 4u * (uint32_t)lg);
    if (EPI == QF_EPI_F32) {
        *(f4*)((float*)p.C + off) = v;
    } else {
        bf4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
        *(bf4*)((bf16_t*)p.C + off) = o;
    }
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); QF_STAMP(p, 3); }
}

// ---- GEMMs whose activations are made on the way in: embedding gather / add + RMSNorm pair (/ + mean pool) ---------
// NCH = H / 256, K = H.  Rows: one WAVE per row, wave w takes rows w, w + 4, ... in batches of RB rows (RB = 1, 2, 4
// by T for T <= 16: one batch; longer queries loop over batches of 4, the next batch's loads issued before the current
// one is reduced).  A batch slot past T re-does row T - 1.
// RO: the K-split reduction buffer `red` OVERLAYS the activation tile (one more barrier between the fragment reads and the
// partial sums) - 48-row blocks with two weight tiles then fit LDS.
template <int NCH, int PRO, int EPI, int NC, int RB, int MT, int NW, int BR = 16 * MT, bool RO = false>
__global__ __launch_bounds__(64 * NW) void qf_gemm_kernel(const QfGemmParams p) {
    constexpr int H = NCH * 256;
    constexpr int NT = EPI == QF_EPI_GEGLU ? 2 : 1;               // weight tiles per workgroup
    constexpr int LDA = H + kQfPad;                                // LDS activation row stride (elements)
    constexpr int GT = PRO == QF_PRO_POOL ? 1 : MT;                // row tiles of the GEMM (POOL: the one pooled row)
    extern __shared__ __attribute__((aligned(16))) unsigned char qf_smem[];
    bf16_t* const sA = (bf16_t*)qf_smem;                           // [16 GT][LDA]
    static_assert(!RO || (size_t)NW * NT * GT * 64 * 16 <= (size_t)16 * GT * LDA * sizeof(bf16_t), "red fits inside the activation tile");
    float* const red = RO ? (float*)qf_smem : (float*)(qf_smem + (size_t)16 * GT * LDA * sizeof(bf16_t));   // [NW waves][NT][GT][64 lanes] f4
    float* const pool = (float*)(qf_smem + (size_t)16 * GT * LDA * sizeof(bf16_t)) + (RO ? 0 : NW * NT * GT * 64 * 4);   // POOL: [NW waves][H] column sums
    bf16_t* const sW = (bf16_t*)(pool + (PRO == QF_PRO_POOL ? NW * H : 0));   // [NT][NC][LDA] weight slice

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    QF_STAMP(p, 0);
    // blockIdx.y = row block (17-32 tokens: two blocks of 16 rows x 16-column tiles - a workgroup then pulls 16 rows of
    // x / y through its CU's L2 port instead of all 32: the prologue is bound by exactly that); rows are
    // local to the block from here on, `T` = how many of them are real
    // (BR < 16: 9-16 tokens as two blocks of 8 rows - half the MFMA tile's rows idle, but 8 rows less to pull and one row
    // per wave in the prologue)
    static_assert(BR <= 16 * MT, "a row block fits the workgroup's row tiles");
    const uint32_t row0 = blockIdx.y * (uint32_t)BR;
    const uint32_t T = p.T - row0 < (uint32_t)BR ? p.T - row0 : (uint32_t)BR;
    constexpr uint32_t kw = H / NW;                                // this wave's K range: [wid kw, (wid + 1) kw)
    constexpr int S = H / NW / 32;                                 // k-steps of 32 per wave
    static_assert(H % (32 * NW) == 0, "the waves split K in whole 32-deep steps");

    // weight rows of the workgroup's tile(s).  GeGLU: W rows are interleaved per 64 (32 gate rows, then the same
    // channels' 32 up rows; embedder.hip set_tensor) -> channels [NC b, NC b + NC) = gate rows 64 (c / 32) + c % 32.
    uint32_t wrow0 = blockIdx.x * (uint32_t)NC;
    if (EPI == QF_EPI_GEGLU) wrow0 = 64u * (wrow0 >> 5) + (wrow0 & 31u);
    // The weight slice (NC consecutive rows per tile = one contiguous block) is requested first, with COALESCED loads
    // (thread = row tid / 32 of the slice, 16-byte chunks tid % 32 + 32 j: a wave-instruction covers 2 rows x 512 B; a
    // fragment gather of 16 rows x 64 B costs ~44 clocks of address processing per instruction against ~16), lands while
    // the rows are normalised and goes through LDS: sW [NT][NC][LDA].
    static_assert(NC == 8 || NC == 16, "an 8- or 16-row weight slice per tile");
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    // slice row i = NC tile + r (i < NT NC) is copied by the 32-thread group i % (2 NW), pass i / (2 NW)
    constexpr int WG32 = 2 * NW;                                   // 32-thread groups of the workgroup
    constexpr int WL = (NT * NC + WG32 - 1) / WG32;                // slice rows per group
    const uint32_t wg = (uint32_t)tid >> 5, wl32 = (uint32_t)tid & 31u;
    u4 wreg[WL][NCH];
#pragma unroll
    for (int t = 0; t < WL; ++t) {
        const uint32_t i = wg + (uint32_t)(WG32 * t);
        const uint32_t ic = i < (uint32_t)(NT * NC) ? i : (uint32_t)(NT * NC - 1);     // (a group past the slice re-copies its last row)
        const bf16_t* src = p.W + (size_t)(wrow0 + 32u * (ic / (uint32_t)NC) + ic % (uint32_t)NC) * H + wl32 * 8u;
#pragma unroll
        for (int j = 0; j < NCH; ++j) wreg[t][j] = *(const u4*)(src + 256 * j);
    }

    // lane owns 4 consecutive floats of each 256-chunk of a row
    f4 wn1[NCH], wp1[NCH];                                         // 1 + w
    const uint32_t c0 = (uint32_t)lane * 4u;
    auto load_row = [&](uint32_t row, f4 (&xv)[NCH], f4 (&yv)[NCH]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (PRO == QF_PRO_EMBED) {
                const bf4 e = *(const bf4*)(p.emb + (size_t)(uint32_t)p.tok[row0 + row] * H + c * 256 + c0);
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[c][i] = (float)e[i] * p.scale;
            } else if (PRO == QF_PRO_ADDLN) {
                const bf4 xb = *(const bf4*)(p.xb_in + (size_t)(row0 + row) * H + c * 256 + c0);
                const bf4 yb = *(const bf4*)(p.y + (size_t)(row0 + row) * H + c * 256 + c0);
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[c][i] = (float)xb[i] + (float)yb[i];
            } else {
                xv[c] = *(const f4*)(p.x_in + (size_t)(row0 + row) * H + c * 256 + c0);
                const bf4 yb = *(const bf4*)(p.y + (size_t)(row0 + row) * H + c * 256 + c0);
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[c][i] = (float)yb[i];
            }
        }
    };
    f4 psum[NCH];                                                  // POOL: this wave's column sums
#pragma unroll
    for (int c = 0; c < NCH; ++c) psum[c] = (f4)(0.f);
    auto finish_row = [&](uint32_t row, bool count, f4 (&xv)[NCH], f4 (&yv)[NCH]) {
        if (PRO == QF_PRO_ADDLN) {
            // LayerNorm of v = x + y (bert_add_ln_kernel's arithmetic: mean, then the biased variance around it, two passes
            // over the registers); wp1 = gamma, wn1 = beta
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) sm += xv[c][i];
            const float mean = qf_wave_sum(sm) / (float)H;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float a = xv[c][i] - mean; q += a * a; }
            const float inv = rsqrtf(qf_wave_sum(q) / (float)H + p.eps);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                bf4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16_t)((xv[c][i] - mean) * inv * wp1[c][i] + wn1[c][i]);
                *(bf4*)(sA + (size_t)row * LDA + c * 256 + c0) = o;
                if (blockIdx.x == 0) *(bf4*)(p.xb_out + (size_t)(row0 + row) * H + c * 256 + c0) = o;
            }
            return;
        }
        if (PRO != QF_PRO_EMBED) {                                   // x += norm(y) (1 + w_post)   (add_norm_kernel's arithmetic)
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) ss += yv[c][i] * yv[c][i];
            const float invy = rsqrtf(qf_wave_sum(ss) / (float)H + p.eps);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[c][i] += yv[c][i] * invy * wp1[c][i];
        }
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) sx += xv[c][i] * xv[c][i];
        const float invx = rsqrtf(qf_wave_sum(sx) / (float)H + p.eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (PRO == QF_PRO_POOL) {
                if (count) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) psum[c][i] += xv[c][i] * invx * wn1[c][i];     // f32 hidden state, summed
                }
            } else {
                bf4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(xv[c][i] * invx * wn1[c][i]);
                *(bf4*)(sA + (size_t)row * LDA + c * 256 + c0) = o;
                if (blockIdx.x == 0) *(f4*)(p.x_out + (size_t)(row0 + row) * H + c * 256 + c0) = xv[c];
            }
        }
    };
    f4 xv[RB][NCH], yv[RB][NCH];
    const uint32_t last = T - 1u;
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const uint32_t r = (uint32_t)wid + (uint32_t)NW * (uint32_t)b;
        load_row(r < T ? r : last, xv[b], yv[b]);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const f4 a = *(const f4*)(p.w_next + c * 256 + c0);
        if (PRO == QF_PRO_ADDLN) {                                   // beta, gamma as they are
            wn1[c] = a;
            wp1[c] = *(const f4*)(p.w_post + c * 256 + c0);
            continue;
        }
        wn1[c] = a + 1.0f;
        if (PRO != QF_PRO_EMBED) wp1[c] = *(const f4*)(p.w_post + c * 256 + c0) + 1.0f;
    }
    if (MT == 1) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const uint32_t r = (uint32_t)wid + (uint32_t)NW * (uint32_t)b;
            finish_row(r < T ? r : last, r < T, xv[b], yv[b]);
        }
    } else {                                                        // RB = 4; 16 rows per pass of the workgroup, passes one after
        for (uint32_t r0 = (uint32_t)wid;;) {                        // the other (a second set of rows in flight spills the registers)
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const uint32_t r = r0 + (uint32_t)NW * (uint32_t)b;
                finish_row(r < T ? r : last, r < T, xv[b], yv[b]);
            }
            r0 += (uint32_t)(NW * RB);
            if (r0 >= T) break;
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const uint32_t r = r0 + (uint32_t)NW * (uint32_t)b;
                load_row(r < T ? r : last, xv[b], yv[b]);
            }
        }
    }
    if (PRO == QF_PRO_POOL) {
        // masked mean pool (src/embedder/pooling.rs:87-128) of the final-norm rows -> ONE activation row (bf16, as
        // mean_pool_kernel hands it to the Dense head)
#pragma unroll
        for (int c = 0; c < NCH; ++c) *(f4*)(pool + (size_t)wid * H + c * 256 + c0) = psum[c];
        __syncthreads();
        for (uint32_t col = (uint32_t)tid; col < (uint32_t)H; col += (uint32_t)(64 * NW)) {
            float sm = pool[col];
#pragma unroll
            for (int w = 1; w < NW; ++w) sm += pool[(size_t)w * H + col];
            sA[col] = (bf16_t)(sm / (float)T);
        }
        // (rows 1..15 of the m-tile stay whatever LDS held: an activation row only feeds its own output column of the
        // MFMA, and columns past the real rows are never stored)
    }
    // rows [T, 16 MT) of the tile are not initialised either, for the same reason
#pragma unroll
    for (int t = 0; t < WL; ++t) {
        const uint32_t i = wg + (uint32_t)(WG32 * t);
        const uint32_t ic = i < (uint32_t)(NT * NC) ? i : (uint32_t)(NT * NC - 1);
#pragma unroll
        for (int j = 0; j < NCH; ++j) *(u4*)(sW + (size_t)ic * LDA + wl32 * 8u + 256 * j) = wreg[t][j];
    }
    __syncthreads();
    QF_STAMP(p, 1);
    // ---- multiply: B operand (activations) from LDS: lane feeds row 16 m + l15, k = wid kw + 32 s + 8 lg .. + 7 ----
    bf8 wf[NT][S];                                                 // weight row l15 % NC (rows past NC repeat real rows)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < S; ++s) wf[t][s] = *(const bf8*)(sW + (size_t)(t * NC + l15 % NC) * LDA + (uint32_t)wid * kw + 8u * (uint32_t)lg + 32 * s);
    f4 acc[NT][GT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < GT; ++m) acc[t][m] = (f4)(0.f);
    const bf16_t* la = sA + (size_t)l15 * LDA + (uint32_t)wid * kw + 8u * (uint32_t)lg;
#pragma unroll
    for (int m = 0; m < GT; ++m)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bf8 a = *(const bf8*)(la + (size_t)(16 * m) * LDA + 32 * s);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][s], a, acc[t][m], 0, 0, 0);
        }
    // ---- sum the four K-quarters through LDS; wave w finishes m-tile w.  acc[t][m][r] = C[row 16 m + l15][col 4 lg + r] ----
    if (RO) __syncthreads();                                        // everybody's fragment reads of the tile `red` overlays are done
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < GT; ++m) *(f4*)(red + ((size_t)((wid * NT + t) * GT + m) * 64 + lane) * 4) = acc[t][m];
    __syncthreads();
    QF_STAMP(p, 2);
    if (wid >= GT) return;
    f4 v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        v[t] = *(const f4*)(red + ((size_t)((0 * NT + t) * GT + wid) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) v[t] += *(const f4*)(red + ((size_t)((w * NT + t) * GT + wid) * 64 + lane) * 4);
    }
    const uint32_t rows = PRO == QF_PRO_POOL ? 1u : T;
    const uint32_t row = 16u * (uint32_t)wid + (uint32_t)l15;
    if (row >= rows || 4 * lg >= NC) return;                       // columns 4 lg .. 4 lg + 3 of the tile: real iff < NC
    const size_t off = (size_t)(row0 + row) * p.ldc + blockIdx.x * (uint32_t)NC + 4u * (uint32_t)lg;
    bf4 o;
    if (EPI == QF_EPI_GEGLU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(qf_gelu_tanh(v[0][r]) * v[NT - 1][r]);
    } else {
        if (PRO == QF_PRO_ADDLN) v[0] = qf_bias_act(v[0], p.bias, p.act, blockIdx.x * (uint32_t)NC + 4u * (uint32_t)lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[0][r];
    }
    *(bf4*)((bf16_t*)p.C + off) = o;
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); QF_STAMP(p, 3); }
}

// ---- attention over <= 64 keys ------------------------------------------------------------------------------------------
// Staging (one WAVE per token row; lane owns dims [4 lane, 4 lane + 4), rotate_half partner = lane ^ 32 - the arithmetic
// of qk_norm_rope_block): per row the k head and the NH q heads are RMS-normalised, rotated (q also scaled) and written
// to LDS as bf16 rows, V is written transposed.  S^T = K Q^T with keys on MFMA rows (softmax lane-local + two
// cross-lane steps), O^T = V^T P^T with the S^T accumulators as the B operand (embed_kernels.hip's key permutation: lane
// group g holds keys {4g..4g+3} of each 16-key tile, so the 8 slots of a 32-key step are keys {4g.., 16 + 4g..} and V^T
// is read in that order).  Two users: qf_attention_kernel (one workgroup per q head; queries of 49-64 tokens) and
// qf_attn_oproj_kernel (every o_proj workgroup redoes the attention of all heads in its own LDS: one launch and one
// round trip through memory less per layer; the redundant work is ~100 MFMAs per workgroup at 8-16 tokens).
constexpr int kQfHD = 256;
constexpr int kQfKRow = kQfHD + 8;          // sQ / sK row stride (elements)

struct QfAttnParams {
    const bf16_t* qkv;        // [64, (heads + 2 kv) 256] bf16, as the QKV projection wrote it
    bf16_t* out;              // qf_attention_kernel: [64, heads 256] bf16
    const float* wq;          // q-head norm weight [256]
    const float* wk;          // k-head norm weight [256]
    const float* cos_sin;     // [max_seq][128][2] of the layer type
    float eps, q_scale;
    uint32_t heads, kv_heads;
    uint32_t window;          // 0 = full attention; else |q - k| < window
    uint32_t T;               // launch parameter (see QfGemmParams)
    const int32_t* pos;       // 65-128 tokens: [128] = 0, 1, 2, ... (the positions qk_norm_rope_kernel rotates by)
    // qf_attn_oproj_kernel only: y = attn Wo^T
    const bf16_t* wo;         // [H, heads 256] bf16
    bf16_t* y;                // [64, H] bf16
    uint32_t H;
    unsigned long long* dbg;
    uint32_t dbg_slot;
};

template <int MT> constexpr int qf_vrow() { return 32 * ((MT + 1) / 2) + 8; }   // sVt row stride: keys padded to 32, + 8

// Stage rows [0, T) of kv head g and q heads [h0, h0 + NH): sQ [NH][16 MTQ][kQfKRow], sK [16 MT][kQfKRow], sVt [256][vrow].
// MTQ < MT: only the query rows [qrow0, qrow0 + 16 MTQ) are staged (a workgroup that owns one row block of the queries
// still needs every key / value row); a row's q heads are then neither loaded nor rotated outside that window.
template <int MT, int RB, int NH, int NW, int MTQ = MT, int QB = 16 * MTQ>
__device__ __forceinline__ void qf_attn_stage(const QfAttnParams& p, uint32_t h0, uint32_t g, bf16_t* sQ, bf16_t* sK, bf16_t* sVt,
                                              int wid, int lane, uint32_t qrow0 = 0u) {
    constexpr int VR = qf_vrow<MT>();
    auto inq = [&](uint32_t row) { return (MTQ == MT && QB == 16 * MTQ) || (row >= qrow0 && row < qrow0 + (uint32_t)QB); };   // (wave-uniform)
    const uint32_t T = p.T, last = T - 1u;
    const uint32_t ld = (p.heads + 2u * p.kv_heads) * (uint32_t)kQfHD;
    const bf16_t* qb = p.qkv + (size_t)h0 * kQfHD + lane * 4;
    const bf16_t* kb = p.qkv + (size_t)(p.heads + g) * kQfHD + lane * 4;
    const bf16_t* vb = p.qkv + (size_t)(p.heads + p.kv_heads + g) * kQfHD + lane * 4;
    const float* csb = p.cos_sin + (uint32_t)(lane & 31) * 8u;
    struct Row { bf4 q[NH], k, v; f4 c0, c1; };
    auto load = [&](uint32_t row, Row& r) {
        if (inq(row)) {
#pragma unroll
            for (int h = 0; h < NH; ++h) r.q[h] = *(const bf4*)(qb + (size_t)row * ld + h * kQfHD);
        }
        r.k = *(const bf4*)(kb + (size_t)row * ld);
        r.v = *(const bf4*)(vb + (size_t)row * ld);
        r.c0 = *(const f4*)(csb + (size_t)row * 256u);
        r.c1 = *(const f4*)(csb + (size_t)row * 256u + 4);
    };
    f4 wq1, wk1;
    auto rot = [&](const bf4& in, const f4& w1, const Row& r, float scale) -> bf4 {
        const float c4[4] = {r.c0[0], r.c0[2], r.c1[0], r.c1[2]};
        const float s4[4] = {r.c0[1], r.c0[3], r.c1[1], r.c1[3]};
        float x[4];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = (float)in[i]; ss += x[i] * x[i]; }
        const float inv = rsqrtf(qf_wave_sum(ss) / (float)kQfHD + p.eps);
        bf4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float n = x[i] * inv * w1[i];
            const float other = qf_xor32(n, lane);                    // rotate_half partner: dim +/- 128
            o[i] = (bf16_t)(((lane < 32) ? (n * c4[i] - other * s4[i]) : (n * c4[i] + other * s4[i])) * scale);
        }
        return o;
    };
    auto put = [&](uint32_t row, const Row& r) {
        if (inq(row)) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
                *(bf4*)(sQ + ((size_t)h * 16 * MTQ + (row - qrow0)) * kQfKRow + lane * 4) = rot(r.q[h], wq1, r, p.q_scale);
        }
        *(bf4*)(sK + (size_t)row * kQfKRow + lane * 4) = rot(r.k, wk1, r, 1.0f);
#pragma unroll
        for (int i = 0; i < 4; ++i) sVt[(size_t)(lane * 4 + i) * VR + row] = r.v[i];
    };
    Row rows[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const uint32_t r = (uint32_t)wid + (uint32_t)NW * (uint32_t)b;
        load(r < T ? r : last, rows[b]);                              // a slot past T re-does row T - 1
    }
    wq1 = *(const f4*)(p.wq + lane * 4) + 1.0f;
    wk1 = *(const f4*)(p.wk + lane * 4) + 1.0f;
    if (MT == 1) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const uint32_t r = (uint32_t)wid + (uint32_t)NW * (uint32_t)b;
            put(r < T ? r : last, rows[b]);
        }
    } else {                                                          // RB = 4: 16 rows per pass of the workgroup
        for (uint32_t r0 = (uint32_t)wid;;) {
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const uint32_t r = r0 + (uint32_t)NW * (uint32_t)b;
                put(r < T ? r : last, rows[b]);
            }
            r0 += (uint32_t)(NW * RB);
            if (r0 >= T) break;
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const uint32_t r = r0 + (uint32_t)NW * (uint32_t)b;
                load(r < T ? r : last, rows[b]);
            }
        }
    }
    // V^T columns [T, 32 ceil(MT / 2)) multiply P = 0 in the last 32-key step: zeros (K rows past T may hold anything:
    // their scores are replaced, not multiplied).  Thread = one head dim.
    if (threadIdx.x < (uint32_t)kQfHD) {
        bf16_t* vz = sVt + (size_t)threadIdx.x * VR;
        for (uint32_t key = T; key < 32u * ((MT + 1) / 2); ++key) vz[key] = (bf16_t)0.f;
    }
}

// One (q head, 16-query tile) unit on one wave: o[j][r] = O[query 16 qt + l15][dim 16 (dt0 + j) + 4 lg + r], unnormalised;
// returns 1 / sum.  sQh = the head's Q rows.
template <int MT, int NDT>
__device__ __forceinline__ float qf_attn_unit(const QfAttnParams& p, const bf16_t* sQh, const bf16_t* sK, const bf16_t* sVt,
                                              uint32_t qt, uint32_t dt0, int l15, int lg, f4 (&o)[NDT], uint32_t qrow0 = 0u) {
    constexpr int VR = qf_vrow<MT>();
    const uint32_t T = p.T;
    const uint32_t ql = 16u * qt + (uint32_t)l15;              // row of sQh (local to the staged query window)
    const uint32_t q = qrow0 + ql;                             // the query's position
    f4 sc[MT];
#pragma unroll
    for (int kt = 0; kt < MT; ++kt) sc[kt] = (f4)(0.f);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const bf8 qf = *(const bf8*)(sQh + (size_t)ql * kQfKRow + 32 * s + 8 * lg);
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const bf8 kf = *(const bf8*)(sK + (size_t)(16 * kt + l15) * kQfKRow + 32 * s + 8 * lg);
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, sc[kt], 0, 0, 0);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t key = (uint32_t)(16 * kt + 4 * lg + r);
            const uint32_t dist = key > q ? key - q : q - key;
            const bool ok = key < T && (p.window == 0u || dist < p.window);
            sc[kt][r] = ok ? sc[kt][r] : -INFINITY;
            mx = fmaxf(mx, sc[kt][r]);
        }
    mx = qf_xor16_max(mx);
    mx = qf_xor32_max(mx);
    float sum = 0.f;
    constexpr int KT2 = 2 * ((MT + 1) / 2);
    bf4 pb[KT2];
#pragma unroll
    for (int kt = 0; kt < KT2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = kt < MT ? __expf(sc[kt < MT ? kt : 0][r] - mx) : 0.f;   // masked: exp(-inf) = 0 (a live query sees itself: mx is finite)
            pb[kt][r] = (bf16_t)e;
            sum += (float)pb[kt][r];                                   // the rounded weights are what multiplies V
        }
    sum = qf_xor16_sum(sum);
    sum = qf_xor32_sum(sum);
#pragma unroll
    for (int j = 0; j < NDT; ++j) o[j] = (f4)(0.f);
#pragma unroll
    for (int kb = 0; kb < (MT + 1) / 2; ++kb) {
        bf8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = pb[2 * kb][r]; pf[4 + r] = pb[2 * kb + 1][r]; }
#pragma unroll
        for (int j = 0; j < NDT; ++j) {
            const bf16_t* vr = sVt + (size_t)(16u * (dt0 + (uint32_t)j) + (uint32_t)l15) * VR + 32 * kb + 4 * lg;
            const bf4 v0 = *(const bf4*)vr, v1 = *(const bf4*)(vr + 16);
            bf8 vf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { vf[r] = v0[r]; vf[4 + r] = v1[r]; }
            o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[j], 0, 0, 0);
        }
    }
    return 1.0f / sum;
}

template <int MT> constexpr size_t qf_attn_lds(int nh) {
    return ((size_t)(nh + 1) * 16 * MT * kQfKRow + (size_t)kQfHD * qf_vrow<MT>()) * sizeof(bf16_t);
}

// Standalone: one workgroup per q head.  Up to 16 queries the 4 waves split the 16 output tiles of the one query tile
// (each recomputes S and the softmax: 8 MFMAs); 17-32 queries: 2 waves per query tile; more: one wave per tile.
template <int MT, int RB>
__global__ __launch_bounds__(256) void qf_attention_kernel(const QfAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qf_smem[];
    bf16_t* const sQ = (bf16_t*)qf_smem;
    bf16_t* const sK = sQ + (size_t)16 * MT * kQfKRow;
    bf16_t* const sVt = sK + (size_t)16 * MT * kQfKRow;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    QF_STAMP(p, 0);
    const uint32_t h = blockIdx.x, g = h / (p.heads / p.kv_heads);
    qf_attn_stage<MT, RB, 1, 4>(p, h, g, sQ, sK, sVt, wid, lane);
    __syncthreads();
    QF_STAMP(p, 1);
    constexpr int G = MT == 1 ? 4 : (MT == 2 ? 2 : 1);                    // waves per query tile
    constexpr int NDT = 16 / G;
    const uint32_t qt = (uint32_t)wid / G, dgrp = (uint32_t)wid % G;
    if (qt >= (uint32_t)MT) return;
    f4 o[NDT];
    const float rinv = qf_attn_unit<MT, NDT>(p, sQ, sK, sVt, qt, dgrp * (uint32_t)NDT, l15, lg, o);
    QF_STAMP(p, 2);
    const uint32_t q = 16u * qt + (uint32_t)l15;
    if (q >= p.T) return;
    bf16_t* orow = p.out + (size_t)q * (p.heads * (uint32_t)kQfHD) + (size_t)h * kQfHD + 4 * lg + 16u * dgrp * (uint32_t)NDT;
#pragma unroll
    for (int j = 0; j < NDT; ++j) {
        bf4 ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)(o[j][r] * rinv);
        *(bf4*)(orow + 16 * j) = ob;
    }
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); QF_STAMP(p, 3); }
}

// Fused: attention of ALL heads (NH = heads, one kv head) in the workgroup's LDS, then y[:, NC columns] = attn Wo^T.
// The attention output of (head, query tile) overwrites that tile's Q rows (the unit's own wave is their only reader).
// MTQ < MT (17-32 tokens: MT = 2 key tiles, MTQ = 1): blockIdx.y = the workgroup's block of 16 MTQ query rows; it stages
// every key / value row but only its own queries (and their cos / sin rows), runs NH MTQ attention units and multiplies
// 16 MTQ rows of o_proj against an NC = 16 column slice - less to pull through the CU's L2 port, half the units per workgroup.
template <int MT, int RB, int NH, int NC, int NW, int MTQ = MT, int QB = 16 * MTQ>
__global__ __launch_bounds__(64 * NW) void qf_attn_oproj_kernel(const QfAttnParams p) {
    constexpr int K = NH * kQfHD, kw = K / NW, S = kw / 32;          // o_proj's K = heads x 256; this wave's K range and k-steps
    static_assert(K % (32 * NW) == 0, "the waves split K in whole 32-deep steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char qf_smem[];
    bf16_t* const sQ = (bf16_t*)qf_smem;                               // [NH][16 MTQ][kQfKRow]: Q, then O
    bf16_t* const sK = sQ + (size_t)NH * 16 * MTQ * kQfKRow;
    bf16_t* const sVt = sK + (size_t)16 * MT * kQfKRow;
    float* const red = (float*)(sVt + (size_t)kQfHD * qf_vrow<MT>());  // [NW waves][MTQ][64 lanes] f4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    QF_STAMP(p, 0);
    static_assert(QB <= 16 * MTQ, "a query block fits the staged query tiles");
    const uint32_t qrow0 = blockIdx.y * (uint32_t)QB;               // (QB < 16: 9-16 tokens as two blocks of 8 queries)
    // o_proj's weight slice (NC consecutive rows: one contiguous block) first, by coalesced loads (thread = row tid / 32,
    // 16-byte chunks tid % 32 + 32 j); it lands while the attention runs and goes through LDS (sW [NC][K + 8]) - except
    // at 3 row tiles, where LDS is full and the fragments are gathered from global memory as before
    static_assert(NC == 8 || NC == 16, "an 8- or 16-row weight slice per workgroup");
    static_assert(NC <= 2 * NW, "one slice row per 32-thread group");
    constexpr bool kStageW = MT < 3 || MTQ < MT;                     // (a block of the queries leaves LDS room at any key count)
    constexpr int LDW = K + 8;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    bf16_t* const sW = (bf16_t*)(red + NW * MTQ * 64 * 4);
    const uint32_t wg = (uint32_t)tid >> 5, wl32 = (uint32_t)tid & 31u;
    u4 wreg[NH];
    bf8 wf[S];
    if (kStageW) {
        if (wg < (uint32_t)NC) {
            const bf16_t* src = p.wo + (size_t)(blockIdx.x * (uint32_t)NC + wg) * K + wl32 * 8u;
#pragma unroll
            for (int j = 0; j < NH; ++j) wreg[j] = *(const u4*)(src + 256 * j);
        }
    } else {
        const bf16_t* wp = p.wo + (size_t)(blockIdx.x * (uint32_t)NC + (uint32_t)(l15 % NC)) * K + wid * kw + 8 * lg;
#pragma unroll
        for (int s = 0; s < S; ++s) wf[s] = *(const bf8*)(wp + 32 * s);
    }
    qf_attn_stage<MT, RB, NH, NW, MTQ, QB>(p, 0u, 0u, sQ, sK, sVt, wid, lane, qrow0);
    __syncthreads();
    QF_STAMP(p, 1);
    for (uint32_t u = (uint32_t)wid; u < (uint32_t)(NH * MTQ); u += (uint32_t)NW) {   // (head, query tile) units over the waves
        const uint32_t h = u / (uint32_t)MTQ, qt = u % (uint32_t)MTQ;
        bf16_t* sQh = sQ + (size_t)h * 16 * MTQ * kQfKRow;
        f4 o[16];
        const float rinv = qf_attn_unit<MT, 16>(p, sQh, sK, sVt, qt, 0u, l15, lg, o, qrow0);
        bf16_t* orow = sQh + (size_t)(16u * qt + (uint32_t)l15) * kQfKRow + 4 * lg;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            bf4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)(o[j][r] * rinv);
            *(bf4*)(orow + 16 * j) = ob;
        }
    }
    if (kStageW && wg < (uint32_t)NC) {
#pragma unroll
        for (int j = 0; j < NH; ++j) *(u4*)(sW + (size_t)wg * LDW + wl32 * 8u + 256 * j) = wreg[j];
    }
    __syncthreads();
    QF_STAMP(p, 2);
    if (kStageW) {
#pragma unroll
        for (int s = 0; s < S; ++s) wf[s] = *(const bf8*)(sW + (size_t)(l15 % NC) * LDW + wid * kw + 8 * lg + 32 * s);
    }
    // y tile: B operand (attention rows) from LDS: row 16 m + l15, k = wid kw + 32 s + 8 lg -> head k / 256, dim k % 256
    f4 acc[MTQ];
#pragma unroll
    for (int m = 0; m < MTQ; ++m) acc[m] = (f4)(0.f);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = wid * kw + 32 * s;                               // wave-uniform; a 32-wide step never straddles a head
        const bf16_t* src = sQ + (size_t)(k / kQfHD) * 16 * MTQ * kQfKRow + (k % kQfHD) + 8 * lg;
#pragma unroll
        for (int m = 0; m < MTQ; ++m) {
            const bf8 a = *(const bf8*)(src + (size_t)(16 * m + l15) * kQfKRow);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], a, acc[m], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < MTQ; ++m) *(f4*)(red + ((size_t)(wid * MTQ + m) * 64 + lane) * 4) = acc[m];
    __syncthreads();
    if (wid >= MTQ) return;
    f4 v = *(const f4*)(red + ((size_t)wid * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < NW; ++w) v += *(const f4*)(red + ((size_t)(w * MTQ + wid) * 64 + lane) * 4);
    const uint32_t lrow = 16u * (uint32_t)wid + (uint32_t)l15, row = qrow0 + lrow;
    if (lrow >= (uint32_t)QB || row >= p.T || 4 * lg >= NC) return;
    bf4 ob;
#pragma unroll
    for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)v[r];
    *(bf4*)(p.y + (size_t)row * p.H + blockIdx.x * (uint32_t)NC + 4u * (uint32_t)lg) = ob;
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); QF_STAMP(p, 3); }
}

// ---- attention + o_proj for 65-128 tokens: the keys in TWO halves ----------------------------------------------------------
// At > 64 keys the one-shot layout above does not fit LDS (K rows + V^T + Q of three heads + the o_proj slice: 195 KB at 128
// keys), and 48 column tiles x 8 query blocks = 384 workgroups of > 80 KB are two rounds on 256 CUs.  Here:
//   * q and k heads are normalised / rotated ONCE per layer by a launch of their own (qk_norm_rope_kernel, in place on the
//     qkv rows: 128 rows x 4 heads) instead of by every workgroup (each pulled 128 KB of cos / sin rows for that, as much as
//     K and V together); the workgroups below only copy rows;
//   * a workgroup = 16 queries x a 32-column o_proj tile (24 x 8 = 192 workgroups: one round), 8 waves; it stages the keys
//     in two halves of 16 MT rows and runs an online softmax across them (running maximum m, per-lane partial sum l, O
//     rescaled by exp(m_old - m_new) between the halves): LDS = what 16 MT keys need (154 KB at MT = 4);
//   * one (head, query tile) unit per wave (waves 0 .. NH - 1); its O accumulators live in registers across the second staging.
template <int MT, int NH, int NW>
__global__ __launch_bounds__(64 * NW) void qf_attn_oproj_long_kernel(const QfAttnParams p) {
    constexpr int NC = 32;                                            // o_proj columns per workgroup = two 16-row weight tiles
    constexpr int K = NH * kQfHD, kw = K / NW, S = kw / 32;
    constexpr int VR = qf_vrow<MT>();
    constexpr int LDW = K + 8;
    static_assert(K % (32 * NW) == 0 && NH <= NW && NC == 4 * NW, "two o_proj weight rows per 32-thread group");
    extern __shared__ __attribute__((aligned(16))) unsigned char qf_smem[];
    bf16_t* const sQ = (bf16_t*)qf_smem;                               // [NH][16][kQfKRow]: Q, then O
    bf16_t* const sK = sQ + (size_t)NH * 16 * kQfKRow;                 // [16 MT][kQfKRow]
    bf16_t* const sVt = sK + (size_t)16 * MT * kQfKRow;                // [256][VR]
    float* const red = (float*)(sVt + (size_t)kQfHD * VR);            // [NW][2][64] f4
    bf16_t* const sW = (bf16_t*)(red + NW * 2 * 64 * 4);               // [NC][LDW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    QF_STAMP(p, 0);
    const uint32_t T = p.T, last = T - 1u;
    const uint32_t qrow0 = blockIdx.y * 16u;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint32_t wg = (uint32_t)tid >> 5, wl32 = (uint32_t)tid & 31u;   // 16 groups of 32 threads: slice rows wg and wg + 16
    u4 wreg[2][NH];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const bf16_t* src = p.wo + (size_t)(blockIdx.x * (uint32_t)NC + wg + 16u * (uint32_t)t) * K + wl32 * 8u;
#pragma unroll
        for (int j = 0; j < NH; ++j) wreg[t][j] = *(const u4*)(src + 256 * j);
    }
    const uint32_t ld = (p.heads + 2u * p.kv_heads) * (uint32_t)kQfHD;
    // ---- Q of the workgroup's 16 queries (already normalised, rotated, scaled): row lr = tid / 32, 16-byte chunks tid % 32 ----
    {
        const uint32_t lr = wg;
        const uint32_t row = qrow0 + lr < T ? qrow0 + lr : last;
        u4 q[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) q[h] = *(const u4*)(p.qkv + (size_t)row * ld + (size_t)h * kQfHD + wl32 * 8u);
#pragma unroll
        for (int h = 0; h < NH; ++h) *(u4*)(sQ + ((size_t)h * 16 + lr) * kQfKRow + wl32 * 8u) = q[h];
    }
    // ---- the two key halves ----
    f4 o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = (f4)(0.f);
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t ql = (uint32_t)l15, qpos = qrow0 + ql;             // this lane's query (row of the staged window / position)
    const bf16_t* const kb = p.qkv + (size_t)p.heads * kQfHD;                          // (one kv head)
    const bf16_t* const vb = p.qkv + (size_t)(p.heads + p.kv_heads) * kQfHD + lane * 4;
    // K rows: plain copies, 16 rows per pass (thread = row tid / 32, chunk tid % 32); V rows: one wave per row, written
    // transposed.  The second half's rows are requested BEFORE the first half's units run (32 registers): their latency
    // hides behind the units instead of following the barrier.
    constexpr int KP = MT;                                            // passes of 16 rows
    constexpr int VP = 16 * MT / NW;                                  // V rows per wave
    u4 kk[KP];
    bf4 vv[VP];
    auto fetch = [&](uint32_t key0, uint32_t kend) {
#pragma unroll
        for (int b = 0; b < KP; ++b) {
            const uint32_t r = key0 + wg + 16u * (uint32_t)b;
            kk[b] = *(const u4*)(kb + (size_t)(r < kend ? r : kend - 1u) * ld + wl32 * 8u);   // a slot past the half re-does its last row
        }
#pragma unroll
        for (int b = 0; b < VP; ++b) {
            const uint32_t r = key0 + (uint32_t)wid + (uint32_t)NW * (uint32_t)b;
            vv[b] = *(const bf4*)(vb + (size_t)(r < kend ? r : kend - 1u) * ld);
        }
    };
    fetch(0u, T < (uint32_t)(16 * MT) ? T : (uint32_t)(16 * MT));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const uint32_t key0 = (uint32_t)(half * 16 * MT);
        const uint32_t kend = T < key0 + (uint32_t)(16 * MT) ? T : key0 + (uint32_t)(16 * MT);   // keys [key0, kend) (T > key0 by the launcher)
        if (half) __syncthreads();                                    // every unit is done with the first half's K / V^T
#pragma unroll
        for (int b = 0; b < KP; ++b) {
            const uint32_t r = key0 + wg + 16u * (uint32_t)b;
            *(u4*)(sK + (size_t)((r < kend ? r : kend - 1u) - key0) * kQfKRow + wl32 * 8u) = kk[b];
        }
#pragma unroll
        for (int b = 0; b < VP; ++b) {
            const uint32_t r = key0 + (uint32_t)wid + (uint32_t)NW * (uint32_t)b;
            const uint32_t lrow = (r < kend ? r : kend - 1u) - key0;
#pragma unroll
            for (int i = 0; i < 4; ++i) sVt[(size_t)(lane * 4 + i) * VR + lrow] = vv[b][i];
        }
        if (half == 0) fetch((uint32_t)(16 * MT), T < (uint32_t)(32 * MT) ? T : (uint32_t)(32 * MT));
        // V^T columns past the half's last key multiply P = 0: zeros (K rows past it hold anything: their scores are replaced)
        if (threadIdx.x < (uint32_t)kQfHD) {
            bf16_t* vz = sVt + (size_t)threadIdx.x * VR;
            for (uint32_t key = kend - key0; key < 32u * ((MT + 1) / 2); ++key) vz[key] = (bf16_t)0.f;
        }
        __syncthreads();
        if (wid < NH) {                                               // unit = head `wid`, the workgroup's one query tile
            const bf16_t* sQh = sQ + (size_t)wid * 16 * kQfKRow;
            f4 sc[MT];
#pragma unroll
            for (int kt = 0; kt < MT; ++kt) sc[kt] = (f4)(0.f);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bf8 qf = *(const bf8*)(sQh + (size_t)ql * kQfKRow + 32 * s + 8 * lg);
#pragma unroll
                for (int kt = 0; kt < MT; ++kt) {
                    const bf8 kf = *(const bf8*)(sK + (size_t)(16 * kt + l15) * kQfKRow + 32 * s + 8 * lg);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, sc[kt], 0, 0, 0);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < MT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t key = key0 + (uint32_t)(16 * kt + 4 * lg + r);
                    const uint32_t dist = key > qpos ? key - qpos : qpos - key;
                    const bool ok = key < T && (p.window == 0u || dist < p.window);
                    sc[kt][r] = ok ? sc[kt][r] : -INFINITY;
                    mx = fmaxf(mx, sc[kt][r]);
                }
            mx = qf_xor32_max(qf_xor16_max(mx));
            const float m_new = fmaxf(m_run, mx);
            const float mb = m_new == -INFINITY ? 0.f : m_new;        // no attendable key so far: every weight is exp(-inf) = 0
            if (half) {                                               // O and the row sum of the first half shrink by exp(m_old - m_new)
                const float a = m_run == -INFINITY ? 0.f : __expf(m_run - mb);
                l_run *= a;
#pragma unroll
                for (int j = 0; j < 16; ++j) o[j] *= a;
            }
            m_run = m_new;
            constexpr int KT2 = 2 * ((MT + 1) / 2);
            bf4 pb[KT2];
#pragma unroll
            for (int kt = 0; kt < KT2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = kt < MT ? __expf(sc[kt < MT ? kt : 0][r] - mb) : 0.f;
                    pb[kt][r] = (bf16_t)e;
                    l_run += (float)pb[kt][r];                        // the rounded weights are what multiplies V
                }
#pragma unroll
            for (int kbk = 0; kbk < (MT + 1) / 2; ++kbk) {
                bf8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pf[r] = pb[2 * kbk][r]; pf[4 + r] = pb[2 * kbk + 1][r]; }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bf16_t* vr = sVt + (size_t)(16u * (uint32_t)j + (uint32_t)l15) * VR + 32 * kbk + 4 * lg;
                    const bf4 v0 = *(const bf4*)vr, v1 = *(const bf4*)(vr + 16);
                    bf8 vf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { vf[r] = v0[r]; vf[4 + r] = v1[r]; }
                    o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[j], 0, 0, 0);
                }
            }
        }
    }
    QF_STAMP(p, 1);
    if (wid < NH) {                                                   // O / sum over the head's Q rows (this wave is their only reader)
        const float sum = qf_xor32_sum(qf_xor16_sum(l_run));
        const float rinv = 1.0f / sum;                                // (a live query sees itself: sum > 0; rows past T are never stored)
        bf16_t* orow = sQ + ((size_t)wid * 16 + (uint32_t)l15) * kQfKRow + 4 * lg;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            bf4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)(o[j][r] * rinv);
            *(bf4*)(orow + 16 * j) = ob;
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < NH; ++j) *(u4*)(sW + (size_t)(wg + 16u * (uint32_t)t) * LDW + wl32 * 8u + 256 * j) = wreg[t][j];
    __syncthreads();
    QF_STAMP(p, 2);
    // ---- y tile = attention rows x the 32-column o_proj slice (two weight tiles); the waves split K ----
    f4 acc[2];
    acc[0] = (f4)(0.f); acc[1] = (f4)(0.f);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = wid * kw + 32 * s;                              // wave-uniform; a 32-wide step never straddles a head
        const bf16_t* src = sQ + (size_t)(k / kQfHD) * 16 * kQfKRow + (k % kQfHD) + 8 * lg;
        const bf8 a = *(const bf8*)(src + (size_t)l15 * kQfKRow);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf8 wf = *(const bf8*)(sW + (size_t)(16 * t + l15) * LDW + k + 8 * lg);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, a, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) *(f4*)(red + ((size_t)(wid * 2 + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    if (wid >= 2) return;                                             // wave t finishes weight tile t
    f4 v = *(const f4*)(red + ((size_t)wid * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < NW; ++w) v += *(const f4*)(red + ((size_t)(w * 2 + wid) * 64 + lane) * 4);
    const uint32_t row = qrow0 + (uint32_t)l15;
    if (row >= T) return;
    bf4 ob;
#pragma unroll
    for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)v[r];
    *(bf4*)(p.y + (size_t)row * p.H + blockIdx.x * (uint32_t)NC + 16u * (uint32_t)wid + 4u * (uint32_t)lg) = ob;
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); QF_STAMP(p, 3); }
}

int qf_debug_repeat() {
    static const int r = [] { const char* e = getenv("CQS_HIP_QUERY_DEBUG_REPEAT"); return e && e[0] == '2' ? 2 : 1; }();
    return r;
}

template <class Kern, class P>
hipError_t qf_launch(Kern kern, DynLdsOnce& once, const P& p0, uint32_t grid_x, size_t lds, hipStream_t st, uint32_t threads = 256u,
                     uint32_t grid_y = 1u) {
    const dim3 grid(grid_x, grid_y);
    const hipError_t e = once.ensure((const void*)kern, lds);
    if (e != hipSuccess) return e;
    if (qf_debug_repeat() == 2 && p0.dbg) {       // diagnostic: every kernel twice (cold vs warm operands), slots 2 s and 2 s + 1
        P q = p0;
        q.dbg_slot = 2 * p0.dbg_slot;
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, q);
        q.dbg_slot++;
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, q);
    } else {
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, p0);
    }
    return hipGetLastError();
}

// (rows per wave in the first batch, row tiles, waves per workgroup) by query length.  Past 8 tokens a workgroup has 8
// waves: the per-row work of the prologue / the attention staging is what these kernels spend their time on.
#define QF_BY_T(T, CALL)                                  \
    do {                                                  \
        if ((T) <= 4u) return CALL(1, 1, 4);              \
        if ((T) <= 8u) return CALL(2, 1, 4);              \
        if ((T) <= 16u) return CALL(2, 1, 8);             \
        if ((T) <= 32u) return CALL(4, 2, 8);             \
        if ((T) <= 48u) return CALL(4, 3, 8);             \
        if ((T) <= 64u) return CALL(4, 4, 8);             \
        return CALL(4, 8, 8);                             \
    } while (0)

template <int NCH, int PRO, int EPI, int NC>
hipError_t qf_launch_pro(const QfGemmParams& p, uint32_t n_out_cols, hipStream_t st) {
    constexpr int NT = EPI == QF_EPI_GEGLU ? 2 : 1;
    // queries over 16 tokens: 8 waves per workgroup (the per-row work of the prologue and of the attention staging is what
    // these kernels spend their time on: half the rows per wave)
#define QF_CALL(RBV, MTV, NWR)                                                                                              \
    [&]() {                                                                                                                 \
        constexpr int GT = PRO == QF_PRO_POOL ? 1 : MTV;                                                                    \
        constexpr int NWV = (NT == 2 && MTV == 4) ? 4 : NWR;               /* (GeGLU x 4 row tiles at 8 waves: LDS) */       \
        const size_t lds = (size_t)(16 * GT + NT * NC) * (NCH * 256 + kQfPad) * sizeof(bf16_t) + (size_t)NWV * NT * GT * 64 * 16 + \
                           (PRO == QF_PRO_POOL ? (size_t)NWV * NCH * 256 * sizeof(float) : 0);                              \
        static DynLdsOnce once;                                                                                             \
        return qf_launch(qf_gemm_kernel<NCH, PRO, EPI, NC, RBV, MTV, NWV>, once, p, n_out_cols / (uint32_t)NC, lds, st,     \
                         64u * NWV);                                                                                        \
    }()
    // over 16 tokens: row blocks of 16 rows x 16-column tiles (a workgroup's prologue pulls 16 rows through its CU's L2
    // port, not all of them, and every column of the MFMA tile is a real one); not for the pooled head (one row tile anyway)
    static const bool split_off = [] { const char* e = getenv("CQS_HIP_QUERY_ROW_SPLIT"); return e && e[0] == '0'; }();
    if (PRO != QF_PRO_POOL && !split_off && p.T > 4u && p.T <= 16u && n_out_cols % 16u == 0) {      // blocks of 8 rows (5-8 tokens: one)
        if (p.T <= 8u) {                     // one block: 8-column tiles (160 / 144 workgroups of 49 / 62 KB beat 80 / 72 of 62 / 86)
            const size_t lds = (size_t)(16 + NT * 8) * (NCH * 256 + kQfPad) * sizeof(bf16_t) + (size_t)8 * NT * 64 * 16;
            static DynLdsOnce once8;
            return qf_launch(qf_gemm_kernel<NCH, PRO == QF_PRO_POOL ? QF_PRO_ADDNORM : PRO, EPI, 8, 1, 1, 8, 8>, once8, p, n_out_cols / 8u, lds, st,
                             512u, (p.T + 7u) / 8u);
        }
        const size_t lds = (size_t)(16 + NT * 16) * (NCH * 256 + kQfPad) * sizeof(bf16_t) + (size_t)8 * NT * 64 * 16;
        static DynLdsOnce once;
        return qf_launch(qf_gemm_kernel<NCH, PRO == QF_PRO_POOL ? QF_PRO_ADDNORM : PRO, EPI, 16, 1, 1, 8, 8>, once, p, n_out_cols / 16u, lds, st,
                         512u, (p.T + 7u) / 8u);
    }
    if (PRO != QF_PRO_POOL && !split_off && p.T > 96u && NT == 2 && n_out_cols % 16u == 0) {
        // GeGLU at 97-128 tokens: row blocks of 48 (72 column tiles x 3 = 216 workgroups: ONE round on 256 CUs; blocks of 32 rows
        // made 288: 12.4 -> 10.7 us).  The QKV projection keeps blocks of 16 rows (48-row blocks: 8.2 -> 9.9 us - its time is the
        // rows a wave normalises, not the rounds).
        constexpr bool RO = NT == 2 && NCH >= 3;                    // (two weight tiles of hidden 768+: `red` overlays the activation tile)
        const size_t lds = (size_t)(48 + NT * 16) * (NCH * 256 + kQfPad) * sizeof(bf16_t) + (RO ? 0 : (size_t)8 * NT * 3 * 64 * 16);
        static DynLdsOnce once48;
        return qf_launch(qf_gemm_kernel<NCH, PRO == QF_PRO_POOL ? QF_PRO_ADDNORM : PRO, EPI, 16, 4, 3, 8, 48, RO>, once48, p, n_out_cols / 16u, lds, st,
                         512u, (p.T + 47u) / 48u);
    }
    if (PRO != QF_PRO_POOL && !split_off && p.T > 64u && NT == 2 && n_out_cols % 16u == 0) {
        // GeGLU over 64 tokens: row blocks of 32 (72 column tiles x 8 blocks of 16 rows = 576 workgroups of 90 KB were 2.25
        // rounds on 256 CUs; x 4 blocks of 32 rows = 288)
        const size_t lds = (size_t)(32 + NT * 16) * (NCH * 256 + kQfPad) * sizeof(bf16_t) + (size_t)8 * NT * 2 * 64 * 16;
        static DynLdsOnce once32;
        return qf_launch(qf_gemm_kernel<NCH, PRO == QF_PRO_POOL ? QF_PRO_ADDNORM : PRO, EPI, 16, 4, 2, 8>, once32, p, n_out_cols / 16u, lds, st,
                         512u, (p.T + 31u) / 32u);
    }
    if (PRO != QF_PRO_POOL && !split_off && p.T > 16u && n_out_cols % 16u == 0) {
        const size_t lds = (size_t)(16 + NT * 16) * (NCH * 256 + kQfPad) * sizeof(bf16_t) + (size_t)8 * NT * 64 * 16;
        static DynLdsOnce once;
        return qf_launch(qf_gemm_kernel<NCH, PRO == QF_PRO_POOL ? QF_PRO_ADDNORM : PRO, EPI, 16, 2, 1, 8>, once, p, n_out_cols / 16u, lds, st,
                         512u, (p.T + 15u) / 16u);
    }
    if (PRO != QF_PRO_POOL && p.T > 64u) return hipErrorNotSupported;     // (only the pooled head keeps all rows in one workgroup)
    QF_BY_T(p.T, QF_CALL);
#undef QF_CALL
}

template <int MT, int EPI, int NC>
hipError_t qf_launch_plain_ch(const QfGemmParams& p, uint32_t n_out_cols, hipStream_t st) {
    const uint32_t steps = p.K / 128u;                               // k-steps of 32 per wave
    const size_t lds = (size_t)4 * MT * 64 * 16;
#define QF_PLAIN(CHV)                                                                                          \
    do {                                                                                                       \
        static DynLdsOnce once;                                                                                \
        return qf_launch(qf_gemm_plain_kernel<MT, CHV, EPI, NC>, once, p, n_out_cols / (uint32_t)NC, lds, st); \
    } while (0)
    if (steps == 9u) QF_PLAIN(9);
    if (steps % 6u == 0u && (MT <= 2 || steps == 6u)) QF_PLAIN(6);   // (batches of 6 x 4 row tiles would not fit the registers twice over)
    if (steps % 4u == 0u) QF_PLAIN(4);
    if (steps % 3u == 0u) QF_PLAIN(3);
    if (steps % 2u == 0u) QF_PLAIN(2);
    QF_PLAIN(1);
#undef QF_PLAIN
}

// K classes of the staged kernel: (k-steps per batch, chunk iterations per row).  Unknown K, 49-64 rows (LDS) or the
// experiment switch CQS_HIP_QUERY_STAGED=0: the gather kernel.
template <int MT, int EPI, int NC, int ONE>
hipError_t qf_launch_staged(const QfGemmParams& p, uint32_t n_out_cols, hipStream_t st, uint32_t row_blocks = 1u) {
    static const bool off = [] { const char* e = getenv("CQS_HIP_QUERY_STAGED"); return e && e[0] == '0'; }();
    if (off || NC != 8) return hipErrorNotSupported;
    const size_t lds = (size_t)(NC + (ONE ? 1 : 16 * MT)) * (p.K + 8u) * sizeof(bf16_t) + (size_t)4 * MT * 64 * 16;
    if (lds > 160u * 1024u) return hipErrorNotSupported;
#define QF_ST(CHV, KCV)                                                                                                  \
    do {                                                                                                                 \
        static DynLdsOnce once;                                                                                          \
        return qf_launch(qf_gemm_staged_kernel<MT, CHV, KCV, EPI, NC, ONE>, once, p, n_out_cols / (uint32_t)NC, lds, st, 256u, row_blocks); \
    } while (0)
    switch (p.K) {
        case 768: QF_ST(6, 3);
        case 1152: QF_ST(9, 5);
        case 3072: QF_ST(8, 12);
        case 512: QF_ST(4, 2);
        case 384: QF_ST(3, 2);
        case 256: QF_ST(2, 1);
        default: return hipErrorNotSupported;
    }
#undef QF_ST
}

template <int EPI, int NC>
hipError_t qf_launch_plain(const QfGemmParams& p, uint32_t n_out_cols, hipStream_t st) {
    const uint32_t rows = p.one_row ? 1u : p.T;
    {
        hipError_t e = hipErrorNotSupported;
        if (p.one_row) e = qf_launch_staged<1, EPI, NC, 1>(p, n_out_cols, st);
        else if (rows <= 16u) e = qf_launch_staged<1, EPI, NC, 0>(p, n_out_cols, st);
        else if (rows <= 32u) e = qf_launch_staged<2, EPI, NC, 0>(p, n_out_cols, st);
        else if (rows <= 48u) e = qf_launch_staged<3, EPI, NC, 0>(p, n_out_cols, st);
        if (e == hipErrorNotSupported && rows > 32u) e = qf_launch_staged<2, EPI, NC, 0>(p, n_out_cols, st, (rows + 31u) / 32u);   // two blocks of 32 rows
        if (e != hipErrorNotSupported) return e;
    }
    if (rows > 64u) return hipErrorNotSupported;                    // (the gather kernel holds at most four row tiles)
    if (rows <= 16u) return qf_launch_plain_ch<1, EPI, NC>(p, n_out_cols, st);
    if (rows <= 32u) return qf_launch_plain_ch<2, EPI, NC>(p, n_out_cols, st);
    if (rows <= 48u) return qf_launch_plain_ch<3, EPI, NC>(p, n_out_cols, st);
    return qf_launch_plain_ch<4, EPI, NC>(p, n_out_cols, st);
}

hipError_t qf_launch_attention(const QfAttnParams& a, hipStream_t st) {       // 4 waves: (rows per wave, row tiles) by length
#define QF_ATT(RBV, MTV)                                                                                   \
    [&]() {                                                                                                \
        static DynLdsOnce once;                                                                            \
        return qf_launch(qf_attention_kernel<MTV, RBV>, once, a, a.heads, qf_attn_lds<MTV>(1), st);        \
    }()
    if (a.T > 64u) return hipErrorNotSupported;
    if (a.T <= 4u) return QF_ATT(1, 1);
    if (a.T <= 8u) return QF_ATT(2, 1);
    if (a.T <= 16u) return QF_ATT(4, 1);
    if (a.T <= 32u) return QF_ATT(4, 2);
    if (a.T <= 48u) return QF_ATT(4, 3);
    return QF_ATT(4, 4);
#undef QF_ATT
}

// attention + o_proj in one launch: heads in {2, 3}, one kv head, <= 48 tokens (LDS); else hipErrorNotSupported
template <int NH, int NC>
hipError_t qf_launch_attn_oproj_h(const QfAttnParams& a, hipStream_t st) {
#define QF_AO(RBV, MTV, NWV)                                                                                                \
    [&]() {                                                                                                                 \
        static DynLdsOnce once;                                                                                             \
        return qf_launch(qf_attn_oproj_kernel<MTV, RBV, NH, NC, NWV>, once, a, a.H / (uint32_t)NC,                          \
                         qf_attn_lds<MTV>(NH) + (size_t)NWV * MTV * 64 * 16 +                                               \
                             (MTV < 3 ? (size_t)NC * (NH * kQfHD + 8) * sizeof(bf16_t) : 0), st, 64u * NWV);                \
    }()
    static const bool mt5_off = [] { const char* e = getenv("CQS_HIP_QUERY_ATTN80"); return e && e[0] == '0'; }();   // A/B hook
    if (a.T > 64u && a.T <= 80u && !mt5_off && a.H % 16u == 0) {
        // 65-80 tokens (round 5): five key tiles still fit the one-shot kernel's LDS next to a 16-query block (154 of 160 KiB
        // with three heads) - no qk_norm_rope launch, no second key half, 4 launches per layer like every shorter query:
        // 0.90 -> 0.78-0.84 ms device.  (Six tiles fit too - 162 304 of 163 840 bytes - but 81-96 tokens are 6 query blocks x 48
        // column tiles = 288 workgroups, two rounds on 256 CUs: 1.09 ms against 0.91 for the two-halves kernel; measured, not kept.)
#define QF_AOS56(MTV)                                                                                                        \
    [&]() -> hipError_t {                                                                                                   \
        static DynLdsOnce once;                                                                                             \
        const size_t lds = ((size_t)(NH * 16 + 16 * MTV) * kQfKRow + (size_t)kQfHD * qf_vrow<MTV>()) * sizeof(bf16_t) +     \
                           (size_t)8 * 64 * 16 + (size_t)16 * (NH * kQfHD + 8) * sizeof(bf16_t);                            \
        if (lds > 160u * 1024u) return hipErrorNotSupported;                                                                \
        return qf_launch(qf_attn_oproj_kernel<MTV, 4, NH, 16, 8, 1, 16>, once, a, a.H / 16u, lds, st, 512u,                 \
                         (a.T + 15u) / 16u);                                                                                \
    }()
        const hipError_t e56 = QF_AOS56(5);
#undef QF_AOS56
        if (e56 != hipErrorNotSupported) return e56;
    }
    if (a.T > 64u) {                                               // 65-128 tokens: the keys in two halves (online softmax across them)
        if (a.H % 32u || a.T > 128u || !a.pos) return hipErrorNotSupported;
        // q / k heads normalised + rotated (q scaled) once, in place on the qkv rows: the workgroups below only copy them
        hipError_t e = launch_qk_norm_rope(const_cast<bf16_t*>(a.qkv), a.pos, a.wq, a.wk, a.cos_sin, a.eps, a.q_scale, a.T, a.heads, a.kv_heads, 0, st);
        if (e != hipSuccess) return e;
#define QF_AOL(MTV)                                                                                                         \
    [&]() {                                                                                                                 \
        static DynLdsOnce once;                                                                                             \
        const size_t lds = ((size_t)(NH * 16 + 16 * MTV) * kQfKRow + (size_t)kQfHD * qf_vrow<MTV>()) * sizeof(bf16_t) +     \
                           (size_t)8 * 2 * 64 * 16 + (size_t)32 * (NH * kQfHD + 8) * sizeof(bf16_t);                        \
        return qf_launch(qf_attn_oproj_long_kernel<MTV, NH, 8>, once, a, a.H / 32u, lds, st, 512u, (a.T + 15u) / 16u);      \
    }()
        if (a.T <= 96u) return QF_AOL(3);                           // two halves of 48 keys
        return QF_AOL(4);                                           // two halves of 64
#undef QF_AOL
    }
    if (a.T <= 4u) return QF_AO(1, 1, 4);
    if (a.T <= 8u) return QF_AO(2, 1, 4);
    static const bool split_off = [] { const char* e = getenv("CQS_HIP_QUERY_ROW_SPLIT"); return e && e[0] == '0'; }();
    if (!split_off && a.H % 16u == 0) {                             // blocks of 8 / 16 queries x 16-column o_proj tiles
#define QF_AOS(MTV, RBV, QBV)                                                                                                \
    [&]() {                                                                                                                 \
        static DynLdsOnce once;                                                                                             \
        const size_t lds = ((size_t)(NH * 16 + 16 * MTV) * kQfKRow + (size_t)kQfHD * qf_vrow<MTV>()) * sizeof(bf16_t) +     \
                           (size_t)8 * 64 * 16 + (size_t)16 * (NH * kQfHD + 8) * sizeof(bf16_t);                            \
        return qf_launch(qf_attn_oproj_kernel<MTV, RBV, NH, 16, 8, 1, QBV>, once, a, a.H / 16u, lds, st, 512u,              \
                         (a.T + (uint32_t)QBV - 1u) / (uint32_t)QBV);                                                       \
    }()
        if (a.T <= 16u) return QF_AOS(1, 2, 8);                     // 9-16 tokens: two blocks of 8 queries
        if (a.T <= 32u) return QF_AOS(2, 4, 16);
        if (a.T <= 48u) return QF_AOS(3, 4, 16);
        return QF_AOS(4, 4, 16);
#undef QF_AOS
    }
    if (a.T <= 16u) return QF_AO(2, 1, 8);
    if (a.T <= 32u) return QF_AO(4, 2, 8);
    if (a.T <= 48u) return QF_AO(4, 3, 8);
    return hipErrorNotSupported;
#undef QF_AO
}
hipError_t qf_launch_attn_oproj(const QfAttnParams& a, hipStream_t st) {
    static const bool off = [] { const char* e = getenv("CQS_HIP_QUERY_FUSE_ATTN"); return e && e[0] == '0'; }();
    if (off || a.kv_heads != 1u || a.H % 8u) return hipErrorNotSupported;
    if (a.heads == 3u) return qf_launch_attn_oproj_h<3, 8>(a, st);
    if (a.heads == 2u) return qf_launch_attn_oproj_h<2, 8>(a, st);
    return hipErrorNotSupported;
}

// Weight-tile width by projection (NC = 8: EmbeddingGemma's QKV 160, o_proj / down 96, GeGLU 144, Dense 1 384 (NC 8),
// Dense 2 96 workgroups): the kernels are instruction-bound, not byte-bound - a narrower tile adds workgroups, not speed.
template <int NCH>
hipError_t qf_forward_t(const QueryFwd& f, hipStream_t st) {
    const uint32_t H = f.hidden, NQ = (f.heads + 2u * f.kv_heads) * 256u, HQ = f.heads * 256u;
    float* xb[2] = {f.x0, f.x1};
    int cur = 0;                                     // the residual stream lives in xb[cur]
    hipError_t e = hipSuccess;
    uint32_t slot = 0;
    for (uint32_t l = 0; l < f.layers; ++l) {
        const QueryFwdLayer& w = f.layer[l];
        QfGemmParams p{};
        p.dbg = f.dbg; p.dbg_slot = slot++;
        p.T = f.T; p.eps = f.eps;
        // QKV (+ embedding gather / the previous layer's post-ffw add + this layer's input norm)
        p.W = w.wqkv; p.C = f.qkv; p.ldc = NQ; p.w_next = w.n_in; p.x_out = xb[cur ^ (l == 0 ? 0 : 1)];
        if (l == 0) {
            p.tok = f.tok; p.emb = f.emb; p.scale = f.embed_scale;
            e = qf_launch_pro<NCH, QF_PRO_EMBED, QF_EPI_BF16, 8>(p, NQ, st);
        } else {
            p.x_in = xb[cur]; p.y = f.y; p.w_post = f.layer[l - 1].n_post_ffw;
            e = qf_launch_pro<NCH, QF_PRO_ADDNORM, QF_EPI_BF16, 8>(p, NQ, st);
            cur ^= 1;
        }
        if (e != hipSuccess) return e;
        // attention
        QfAttnParams a{};
        a.dbg = f.dbg; a.dbg_slot = slot++;
        a.T = f.T; a.qkv = f.qkv; a.out = f.attn; a.wq = w.n_q; a.wk = w.n_k;
        const bool full = ((l + 1u) % f.sliding_pattern) == 0u;
        a.cos_sin = full ? f.rope_global : f.rope_local;
        a.eps = f.eps; a.q_scale = f.q_scale; a.heads = f.heads; a.kv_heads = f.kv_heads; a.window = full ? 0u : f.window;
        a.wo = w.wo; a.y = f.y; a.H = H; a.pos = f.pos;
        e = qf_launch_attn_oproj(a, st);                 // attention + o_proj in one launch where it fits ...
        if (e == hipErrorNotSupported) {                 // ... else two
            if ((e = qf_launch_attention(a, st)) != hipSuccess) return e;
            QfGemmParams po{};
            po.dbg = f.dbg; po.dbg_slot = slot;
            po.T = f.T; po.A = f.attn; po.K = HQ; po.W = w.wo; po.C = f.y; po.ldc = H;
            e = qf_launch_plain<QF_EPI_BF16, 8>(po, H, st);
        }
        slot++;
        if (e != hipSuccess) return e;
        // GeGLU (+ post-attention add, pre-ffw norm)
        QfGemmParams pg{};
        pg.dbg = f.dbg; pg.dbg_slot = slot++;
        pg.T = f.T; pg.eps = f.eps; pg.x_in = xb[cur]; pg.y = f.y; pg.w_post = w.n_post_attn; pg.w_next = w.n_pre_ffw;
        pg.x_out = xb[cur ^ 1]; pg.W = w.wgu; pg.C = f.h; pg.ldc = f.inter;
        if ((e = qf_launch_pro<NCH, QF_PRO_ADDNORM, QF_EPI_GEGLU, 8>(pg, f.inter, st)) != hipSuccess) return e;
        cur ^= 1;
        // down
        QfGemmParams pd{};
        pd.dbg = f.dbg; pd.dbg_slot = slot++;
        pd.T = f.T; pd.A = f.h; pd.K = f.inter; pd.W = w.wd; pd.C = f.y; pd.ldc = H;
        if ((e = qf_launch_plain<QF_EPI_BF16, 8>(pd, H, st)) != hipSuccess) return e;
    }
    // head: last post-ffw add + final norm + mean pool -> Dense 1 ; Dense 2
    QfGemmParams p1{};
    p1.dbg = f.dbg; p1.dbg_slot = slot++;
    p1.T = f.T; p1.eps = f.eps; p1.x_in = xb[cur]; p1.y = f.y; p1.w_post = f.layer[f.layers - 1].n_post_ffw; p1.w_next = f.n_final;
    p1.W = f.dense1; p1.C = f.d1; p1.ldc = f.dense_hidden; p1.one_row = 1;
    if ((e = qf_launch_pro<NCH, QF_PRO_POOL, QF_EPI_BF16, 8>(p1, f.dense_hidden, st)) != hipSuccess) return e;
    QfGemmParams p2{};
    p2.dbg = f.dbg; p2.dbg_slot = slot++;
    p2.T = f.T; p2.A = f.d1; p2.K = f.dense_hidden; p2.W = f.dense2; p2.C = f.out; p2.ldc = H; p2.one_row = 1;
    return qf_launch_plain<QF_EPI_F32, 8>(p2, H, st);
}

}  // namespace

// C[M, N] = act(A[M, K] W[N, K]^T + bias) for M <= 64 rows (a search-time SPLADE query, a rerank of one short passage):
// the search-time GEMM kernels above - 8 output columns per workgroup, K split over its 4 waves, both operands staged
// through LDS by coalesced loads where they fit - instead of one wave walking all of K per 32 x 32 tile.
// hipErrorNotSupported: shape outside what those kernels take (the caller falls back).
hipError_t launch_gemm_small_rows(const bf16_t* A, const bf16_t* W, const float* bias, void* C, uint32_t M, uint32_t N,
                                  uint32_t K, uint32_t ldc, GemmOut out, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (M > 64u || N % 8u || K % 128u || K < 128u) return hipErrorNotSupported;   // (the BERT engines' search-time shapes; the Gemma chain itself goes to 128 rows)
    if (out != GEMM_OUT_BF16 && out != GEMM_OUT_BF16_GELU && out != GEMM_OUT_F32) return hipErrorNotSupported;
    QfGemmParams p{};
    p.T = M; p.A = A; p.K = K; p.W = W; p.C = C; p.ldc = ldc; p.bias = bias; p.act = out == GEMM_OUT_BF16_GELU ? 1 : 0;
    if (out == GEMM_OUT_F32) return qf_launch_plain<QF_EPI_F32, 8>(p, N, st);
    return qf_launch_plain<QF_EPI_BF16, 8>(p, N, st);
}

// The same with the producer's residual add + LayerNorm in the prologue (BERT's post-LN layers):
//   x_out = bf16(LayerNorm(x + y) gamma + beta);  C = act(x_out W^T + bias)
// for M <= 64 rows, K = H in {256, 768, 1024}; x_out != x (every workgroup reads x, column tile 0 writes x_out).
hipError_t launch_gemm_small_rows_addln(const bf16_t* x, const bf16_t* y, const float* gamma, const float* beta, float eps,
                                        bf16_t* x_out, const bf16_t* W, const float* bias, void* C, uint32_t M, uint32_t N,
                                        uint32_t H, uint32_t ldc, GemmOut out, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (M > 64u || N % 16u || x == x_out || (out != GEMM_OUT_BF16 && out != GEMM_OUT_BF16_GELU)) return hipErrorNotSupported;
    QfGemmParams p{};
    p.T = M; p.xb_in = x; p.y = y; p.w_post = gamma; p.w_next = beta; p.eps = eps; p.xb_out = x_out;
    p.W = W; p.C = C; p.ldc = ldc; p.bias = bias; p.act = out == GEMM_OUT_BF16_GELU ? 1 : 0;
    if (H == 768u) return qf_launch_pro<3, QF_PRO_ADDLN, QF_EPI_BF16, 8>(p, N, st);
    if (H == 1024u) return qf_launch_pro<4, QF_PRO_ADDLN, QF_EPI_BF16, 8>(p, N, st);
    if (H == 256u) return qf_launch_pro<1, QF_PRO_ADDLN, QF_EPI_BF16, 8>(p, N, st);
    return hipErrorNotSupported;
}

bool query_forward_supported(const EmbedGeom& g) {
    const uint32_t HQ = g.heads * 256u;
    return (g.hidden == 256u || g.hidden == 768u) && g.head_dim == 256u && g.kv_heads && g.heads % g.kv_heads == 0 &&
           g.inter % 128u == 0 && HQ % 128u == 0 && g.dense_hidden % 128u == 0 && g.hidden % 128u == 0;
}

// Longest query the chain serves for this geometry: 128 tokens where the two-half attention + o_proj kernel and the staged
// GEMMs apply (2 or 3 q heads on one kv head, K in the staged kernel's classes), else 64.
uint32_t query_forward_max_tokens(const EmbedGeom& g) {
    static const bool fuse_off = [] { const char* e = getenv("CQS_HIP_QUERY_FUSE_ATTN"); return e && e[0] == '0'; }();
    static const bool staged_off = [] { const char* e = getenv("CQS_HIP_QUERY_STAGED"); return e && e[0] == '0'; }();
    static const bool split_off = [] { const char* e = getenv("CQS_HIP_QUERY_ROW_SPLIT"); return e && e[0] == '0'; }();
    auto staged_k = [](uint32_t k) { return k == 768u || k == 1152u || k == 3072u || k == 512u || k == 384u || k == 256u; };
    const bool long_ok = !fuse_off && !staged_off && !split_off && g.kv_heads == 1u && (g.heads == 2u || g.heads == 3u) && g.hidden % 16u == 0 &&
                         g.inter % 16u == 0 && staged_k(g.heads * 256u) && staged_k(g.inter);
    return long_ok ? kQueryFwdMaxTokens : 64u;
}

hipError_t launch_query_forward(const QueryFwd& f, hipStream_t st) {
    if (!f.layer || f.layers == 0 || f.T < 1u || f.T > kQueryFwdMaxTokens) return hipErrorInvalidValue;
    if (f.hidden == 768u) return qf_forward_t<3>(f, st);
    if (f.hidden == 256u) return qf_forward_t<1>(f, st);
    return hipErrorInvalidValue;
}

}  // namespace cqs
