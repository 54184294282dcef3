// gemm_kernels.hip — bf16 GEMM for the embedding forward, second generation: C[M,N] = A[M,K] W[N,K]^T on
// 256 x (64 TN) x 64 tiles, 8 waves (2 per SIMD, one workgroup per CU), v_mfma_f32_16x16x32_bf16.
//
// Why a second kernel (gemm_bf16_kernel in embed_kernels.hip stays for small / odd shapes and partly filled rounds):
// a 128 x 128 tile moves 32 KB of operands L2 -> LDS per 64-deep K-step for 512 MFMA cycles per SIMD - 64 B/clk/CU
// against the ~56 B/clk/CU an XCD's L2 delivers - and re-reads the A panel once per 128 output columns (PMC, round 2:
// 169 MB fetched by the GeGLU GEMM against 28 MB of operands).  A 256 x 256 tile needs half of both.
//   * wave grid 2 (M) x 4 (N): a wave owns 128 rows x 16 TN columns = 8 x TN accumulator tiles (128 regs at TN = 4);
//     per 32-deep k-step 8 + TN fragment reads feed 8 TN MFMAs;
//   * operands by LDS-DMA (global_load_lds_dwordx4, 16 B per lane, full 128-B lines, bank swizzle on the SOURCE
//     address), counted `s_waitcnt vmcnt(N)` (never 0 in the main loop) + raw `s_barrier`: the DMA stays in flight
//     across barriers (a __syncthreads() would drain it);
//   * operand roles swapped: W rows are the MFMA's A operand, activation rows its B operand, so a lane ends up
//     with 4 CONSECUTIVE output columns of one row (GeGLU pairs stay in-lane);
//   * PING-PONG time structure (see gemm_pp_kernel) and an LDS-staged epilogue that stores whole rows.
// History (measured on 4096^3, random operands): all waves in lock-step with fragments read in the phase that uses
// them 1058-1150 TF; the same with fragments prefetched one phase ahead 1030-1070 TF (ablations: MFMA alone half the
// time, DMA alone the other half, nothing overlapped because every wave did the same thing at the same time);
// ping-pong 1290-1360 TF.
#include "embed_kernels.h"

#include <utility>

#include <cstdlib>
#include <type_traits>

namespace cqs {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kP8M = 256;                 // tile rows
constexpr int kARegion = 256 * 64;        // elements of the A tile per LDS buffer (32 KB)

// erf GELU (BERT's hidden_act = "gelu"): 0.5 x (1 + erf(x / sqrt 2)), erf by Abramowitz-Stegun 7.1.26 (|error| <=
// 1.5e-7: one exp + one rcp + a degree-5 polynomial; libm's erff costs 40 % of the whole 768 -> 3072 GEMM here)
__device__ __forceinline__ float p8_gelu_erf(float x) {
    const float z = __builtin_fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = 1.0f - poly * __expf(-z * z);            // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + __builtin_copysignf(e, x));
}
// gelu_tanh(g) * u for two channels at once, in packed f32 arithmetic (v_pk_mul / v_pk_fma: two lanes' worth per instruction;
// the epilogue holds no MFMAs for them to disturb): x sigmoid(2 k0 (x + k1 x^3)) = x / (1 + 2^(x (c0 + c1 x^2))),
// c0 = -2 k0 log2(e), c1 = c0 k1: per pair 2 v_exp + 2 v_rcp + 6 packed operations.
typedef float p8_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ p8_f2 p8_geglu2(p8_f2 g, p8_f2 u) {
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
    const p8_f2 t = g * g;
    const p8_f2 w = g * __builtin_elementwise_fma(t, (p8_f2)(c1), (p8_f2)(c0));
    p8_f2 e;
    e[0] = __builtin_amdgcn_exp2f(w[0]);
    e[1] = __builtin_amdgcn_exp2f(w[1]);
    const p8_f2 d = e + (p8_f2)(1.0f);
    p8_f2 r;
    r[0] = __builtin_amdgcn_rcpf(d[0]);
    r[1] = __builtin_amdgcn_rcpf(d[1]);
    return g * r * u;
}
// erf GELU for two values at once in packed f32 arithmetic (same polynomial and operation order per element as p8_gelu_erf)
__device__ __forceinline__ p8_f2 p8_gelu_erf2(p8_f2 x) {
    p8_f2 ax;
    ax[0] = __builtin_fabsf(x[0]); ax[1] = __builtin_fabsf(x[1]);
    const p8_f2 z = ax * (p8_f2)(0.70710678118654752f);
    const p8_f2 d = __builtin_elementwise_fma(z, (p8_f2)(0.3275911f), (p8_f2)(1.0f));
    p8_f2 t;
    t[0] = __frcp_rn(d[0]); t[1] = __frcp_rn(d[1]);
    p8_f2 poly = __builtin_elementwise_fma(t, (p8_f2)(1.061405429f), (p8_f2)(-1.453152027f));
    poly = __builtin_elementwise_fma(poly, t, (p8_f2)(1.421413741f));
    poly = __builtin_elementwise_fma(poly, t, (p8_f2)(-0.284496736f));
    poly = __builtin_elementwise_fma(poly, t, (p8_f2)(0.254829592f));
    poly = poly * t;
    const p8_f2 nz2 = -(z * z);
    p8_f2 ex;
    ex[0] = __expf(nz2[0]); ex[1] = __expf(nz2[1]);
    const p8_f2 e = (p8_f2)(1.0f) - poly * ex;               // erf(|x| / sqrt 2)
    p8_f2 se;
    se[0] = __builtin_copysignf(e[0], x[0]); se[1] = __builtin_copysignf(e[1], x[1]);
    return (p8_f2)(0.5f) * x * ((p8_f2)(1.0f) + se);
}
__device__ __forceinline__ float p8_gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = 2.0f * k0 * (x + k1 * x * x * x);
    return x * __frcp_rn(1.0f + __expf(-u2));
}

#define P8_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// the LDS reads a phase issued have long returned by its end; the explicit wait makes the refill of their slot
// (issued by other waves right after the barrier) safe by construction
#ifdef P8_ABLATE_NOBAR
#define P8_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define P8_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// LDS image of one buffer: [A: 256 rows][B: 64 TN rows], 128 B (64 bf16) per row, 16-B chunk c of row R stored at
// chunk c ^ ((R >> 1) & 7) (conflict-free 16 x 32 fragment reads, see embed_kernels.hip swz()).  Row order:
//   A region row R: piece h = R / 128, m-tile t = (R % 128) / 32, wave row wm = (R / 16) % 2, r = R % 16
//                   -> tile row m = wm * 128 + (4 h + t) * 16 + r
//   B region row R: n-tile t = R / 64 (tiles [0, TN0) = piece B0, the rest B1), wn = (R / 16) % 4, r = R % 16
//                   -> tile column n = wn * 16 TN + t * 16 + r
// ---- the kernel: ping-pong time structure -------------------------------------------------------------------------
// With all eight waves in lock-step (issue DMA, read fragments, wait, multiply - everybody at the same time) the matrix
// pipe idles while a wave does anything else.  Here the workgroup's two wave rows run ONE INTERVAL apart (waves 0-3 start, waves 4-7 take one extra barrier first
// and one fewer at the end): every wave alternates a LOAD interval (fragment reads for its next multiply, DMA refills,
// counted DMA wait) with an MMA interval (32 MFMAs, nothing else), and on every SIMD one wave multiplies while its
// partner loads.  A K-tile is two phases (A half h x the whole B tile); pieces A0 | B | A1:
//   LOAD(kt,0): read A0(kt), B(kt)   | refill A1's slot of buffer kt+1 (last read in phase (kt-1,1)) | wait A1(kt)
//   LOAD(kt,1): read A1(kt)          | refill A0 and B of K-tile kt+2 (read in phase (kt,0))         | wait A0,B(kt+1)
// A piece read by both wave rows is dead one interval after the later row's LOAD; the refill sits one phase later, so
// the same program is hazard-free for both rows.  DMA flight time: two phases (four intervals).
template <int TN, int OUT>
__device__ __forceinline__ void gemm_pp_body(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                             void* __restrict__ Cv, uint32_t M, uint32_t N, uint32_t K, uint32_t ldc,
                                             const uint32_t bid /*workgroup index among this problem's*/,
                                             const float* __restrict__ bias = nullptr /*[N] added before the activation*/,
                                             const int32_t* __restrict__ row_seq = nullptr /*ROWMAX: sequence of each row*/,
                                             uint32_t n_valid = 0 /*ROWMAX: columns >= n_valid are padding*/,
                                             const QkvEpilogue* __restrict__ epi = nullptr /*QKV (kernel argument memory)*/) {
    constexpr int BN = 64 * TN;
    constexpr int kBuf = (kP8M + BN) * 64;                      // elements per LDS buffer
    constexpr int PA = 2, PB = TN;                              // DMA instructions per wave: an A half / the B tile
    extern __shared__ __attribute__((aligned(16))) bf16_t p8smem[];   // [2][A 256 x 64 | B BN x 64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    const uint32_t nt = N / BN, mt = (M + kP8M - 1) / kP8M, total = nt * mt;
    const uint32_t xcd = bid % 8u, q8 = total / 8u, r8 = total % 8u;   // (bid % 8 = the XCD: parts of a dual launch start at multiples of 8)
    const uint32_t tile = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + bid / 8u;
    const uint32_t m0 = (tile / nt) * kP8M, n0 = (tile % nt) * BN;

    // DMA sources.  Instruction i of a piece covers 8 LDS rows: lane -> row (lane >> 3), physical chunk (lane & 7) ->
    // logical chunk (lane & 7) ^ ((R >> 1) & 7).  For the A piece h, instruction j of wave w (i = w + 8 j):
    //   tile row m = ((w >> 1) & 1) * 128 + (4 h + (w >> 2) + 2 j) * 16 + (w & 1) * 8 + (lane >> 3)
    // and for n-tile t of the B tile: column n = (w >> 1) * 16 TN + t * 16 + (w & 1) * 8 + (lane >> 3); the swizzle term
    // depends on (w & 1, lane) only.  So ONE per-lane byte offset per operand lives in a VGPR; the (h, j) / t parts are
    // wave-uniform and go into the scalar base.  In the last, partial M tile the lanes whose row is >= M read row M - 1
    // instead (any finite row: their outputs are never stored).
    const uint32_t rr = (uint32_t)((wid & 1) * 8 + (lane >> 3));
    const uint32_t lcw = (uint32_t)(lane & 7) ^ ((rr >> 1) & 7u);
    const uint32_t offA0 = ((((uint32_t)(wid >> 1) & 1u) * 128u + (uint32_t)(wid >> 2) * 16u + rr) * K + lcw * 8u) * 2u;
    const uint32_t offB0 = (((uint32_t)(wid >> 1) * (uint32_t)(16 * TN) + rr) * K + lcw * 8u) * 2u;
    const size_t rowK = (size_t)K * 32u;        // bytes of 16 rows
    const char* const gA = (const char*)(A + (size_t)m0 * K);
    const char* const gW = (const char*)(W + (size_t)n0 * K);
    // LDS-DMA by hand (global_load_lds_dwordx4 with a scalar base + one 32-bit lane offset): through the builtin hipcc
    // hoists one 64-bit VGPR address per instruction out of the K loop (18-22 VGPRs this kernel does not have).  M0 =
    // LDS byte address of the wave's 1 KiB; nothing else in this kernel uses M0.  The statement is invisible to hipcc's
    // waitcnt bookkeeping, which is what the counted vmcnt scheme wants anyway.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)p8smem;
    auto dma = [&](const char* sbase, uint32_t voff, uint32_t lds_elem) {
        const uint32_t m0v = lds0 + lds_elem * 2u;
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(m0v), "v"(voff), "s"(sbase) : "memory");
    };
    const uint32_t rows_left = M - m0;           // >= 1
    const uint32_t offClamp = ((rows_left - 1u) * K + lcw * 8u) * 2u;
    auto stage_a = [&](int h, uint32_t kt, int par) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const uint32_t lds_elem = (uint32_t)(par * kBuf + (h * 128 + (wid + 8 * j) * 8) * 64);
            const uint32_t row0 = ((uint32_t)(wid >> 1) & 1u) * 128u + (uint32_t)(4 * h + (wid >> 2) + 2 * j) * 16u;   // of rr = 0
            if (row0 + 16u <= rows_left) {           // wave-uniform: the whole instruction is inside the matrix
                dma(gA + ((size_t)kt * 128u + (size_t)(4 * h + 2 * j) * rowK), offA0, lds_elem);
            } else {
                const uint32_t voff = (row0 + rr < rows_left) ? offA0 + (uint32_t)(4 * h + 2 * j) * (uint32_t)rowK : offClamp;
                dma(gA + (size_t)kt * 128u, voff, lds_elem);
            }
        }
    };
    auto stage_b = [&](uint32_t kt, int par) {
#pragma unroll
        for (int t = 0; t < TN; ++t)
            dma(gW + ((size_t)kt * 128u + (size_t)t * rowK), offB0, (uint32_t)(par * kBuf + kARegion + (t * 64 + wid * 8) * 64));
    };

    const uint32_t sw = (uint32_t)(lane >> 1) & 7u;
    const bf16_t* la[2];
    const bf16_t* lb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t f = (uint32_t)l15 * 64u + (((uint32_t)(4 * s + lg)) ^ sw) * 8u;
        la[s] = p8smem + (uint32_t)(wm * 16) * 64u + f;
        lb[s] = p8smem + (uint32_t)kARegion + (uint32_t)(wn * 16) * 64u + f;
    }

    f4 acc[8][TN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f4)(0.f);
    bf8 xa[4][2];           // the current A half: [m-tile][k-sub]
    bf8 wb[TN][2];          // the K-tile's W fragments (read in phase 0, kept for phase 1)

    auto read_a = [&](int par, int h) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) xa[t][s] = *(const bf8*)(la[s] + (par * kBuf + h * 128 * 64 + t * 32 * 64));
    };
    auto read_b = [&](int par) {
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) wb[t][s] = *(const bf8*)(lb[s] + (par * kBuf + t * 64 * 64));
    };
    auto mma = [&](int h) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[4 * h + t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j][s], xa[t][s], acc[4 * h + t][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // DMA pieces in issue order: A0 B A1 of K-tile 0, 1, 2, ... (index 3 kt + pos; PA / PB / PA instructions per wave).
    const uint32_t nk = K / 64u;
    const uint32_t n_pieces = 3u * nk;
    constexpr int kCnt[3] = {PA, PB, PA};
    auto wait_tail = [&](auto pos_c, uint32_t needed_idx) {   // fewer than three younger pieces exist
        constexpr int pos = decltype(pos_c)::value;
        if (needed_idx >= n_pieces) return;
        switch (n_pieces - 1u - needed_idx) {
            case 0: P8_WAIT_VM(0); break;
            case 1: P8_WAIT_VM(kCnt[(pos + 1) % 3]); break;
            case 2: P8_WAIT_VM(kCnt[(pos + 1) % 3] + kCnt[(pos + 2) % 3]); break;
            default: P8_WAIT_VM(PA + PB + PA); break;
        }
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    constexpr int kAll = PA + PB + PA;

    // prologue (all waves together): A0 B A1 of K-tile 0, A0 B of K-tile 1; A0 / B of K-tile 0 must have landed
    stage_a(0, 0u, 0); stage_b(0u, 0); stage_a(1, 0u, 0);
    if (nk > 1u) { stage_a(0, 1u, 1); stage_b(1u, 1); }
    if (nk > 1u) P8_WAIT_VM(kAll); else P8_WAIT_VM(PA);       // younger than B(0): A1(0) | A0(1) B(1)
    P8_BARRIER();
    if (wm == 1) P8_BARRIER();                                 // wave row 1 runs one interval behind

    auto tile_body = [&](uint32_t kt, auto par_c) {
        constexpr int par = decltype(par_c)::value;
        const bool steady = kt + 2u < nk;                      // every piece up to B (kt+2) exists
        // LOAD (kt, 0)
        read_a(par, 0);
        read_b(par);
        if (kt + 1u < nk) stage_a(1, kt + 1u, par ^ 1);
        if (steady) P8_WAIT_VM(kAll); else wait_tail(I2{}, 3u * kt + 2u);        // A1 (kt); younger: A0 B A1 (kt+1)
        P8_BARRIER();
        mma(0);
        P8_BARRIER();
        // LOAD (kt, 1)
        read_a(par, 1);
        if (kt + 2u < nk) { stage_a(0, kt + 2u, par); stage_b(kt + 2u, par); }
        if (steady) P8_WAIT_VM(kAll); else wait_tail(I1{}, 3u * kt + 4u);        // B (kt+1); younger: A1 (kt+1) A0 B (kt+2)
        P8_BARRIER();
        mma(1);
        P8_BARRIER();
    };
    for (uint32_t kt = 0; kt < nk; kt += 2u) {
        tile_body(kt, P0{});
        if (kt + 1u < nk) tile_body(kt + 1u, P1{});
    }
    if (wm == 0) P8_BARRIER();                                 // wave row 0 waits for row 1's last interval

#if defined(P8_ABLATE_NOEPI)     // timing experiment (wrong results): the main loop alone
    {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) sacc += acc[i][j][0] + acc[i][j][3];
        if (sacc == 12345.678f) *(float*)Cv = sacc;
        return;
    }
#endif
    // ---- epilogue: acc[i][j][r] = C[m = wm*128 + i*16 + l15][n = wn*16TN + j*16 + 4 lg + r]
    if (OUT == GEMM_OUT_F32) {           // (only the two tiny Dense GEMMs of the head: direct 16-byte stores)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t row = m0 + (uint32_t)(wm * 128 + i * 16 + l15);
            if (row >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const uint32_t col = n0 + (uint32_t)(wn * 16 * TN + j * 16 + 4 * lg);
                f4 v = acc[i][j];
                if (bias) v += *(const f4*)(bias + col);
                *(f4*)((float*)Cv + (size_t)row * ldc + col) = v;
            }
        }
        return;
    }
    // bf16 / GeGLU: through LDS, so that the workgroup writes WHOLE rows of its tile (a lane's 8 bytes of a 16 x 16
    // accumulator tile are a quarter of a 32-byte segment: stored directly, every 128-byte line is written in four
    // pieces by four instructions - measured: the store phase was 25-35 % of these K = 768 GEMMs).
    if (OUT == GEMM_OUT_ROWMAX) {
        // Column maxima per sequence instead of the tile (SPLADE pooling, src/splade/mod.rs:1026-1043, folded into the
        // decoder GEMM: the [tokens, vocab] logits never reach HBM).  The reference pools F32 logits (max, then
        // ln(1 + max(0, x)), then `> threshold`), so the f32 accumulators (+ bias) are staged as they are - a bf16
        // stage would add a 2^-9 relative rounding on top of the GEMM's own error and flip entries at the threshold.
        // Four passes of 64 rows (m-tiles 2p, 2p+1 of both wave rows), LDS row = BN f32 + 16 bytes.  Thread = one
        // column x one of the pass's two 32-row runs; the running maximum of max(0, x) is flushed with an atomic max
        // on the bits (non-negative floats order like unsigned integers; x <= 0 and NaN never pass `> 0`, which is the
        // reference's strict `>` from -inf followed by max(., 0)) whenever the row's sequence changes.
        f4 bv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bv[j] = bias ? *(const f4*)(bias + n0 + (uint32_t)(wn * 16 * TN + j * 16 + 4 * lg)) : (f4)(0.f);
        constexpr int kStride = BN + 4;                             // floats
        float* const stage = (float*)p8smem;                        // 64 x kStride floats <= 66 KiB
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float* rowp = stage + (size_t)(wm * 32 + t * 16 + l15) * kStride;
#pragma unroll
                for (int j = 0; j < TN; ++j) *(f4*)(rowp + wn * 16 * TN + j * 16 + 4 * lg) = acc[2 * p + t][j] + bv[j];
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const uint32_t col = (uint32_t)tid % (uint32_t)BN, run = (uint32_t)tid / (uint32_t)BN;
                const uint32_t v = n0 + col;
                const uint32_t row0 = m0 + run * 128u + (uint32_t)(2 * p) * 16u;
                if (v < n_valid && row0 < M) {
                    uint32_t* const outp = (uint32_t*)Cv + v;
                    const uint32_t cnt = M - row0 < 32u ? M - row0 : 32u;
                    const float* const sp = stage + (size_t)(run * 32u) * kStride + col;
                    int32_t cur = row_seq[row0];
                    float best = 0.f;
                    if (cur == row_seq[row0 + cnt - 1u]) {             // rows are in sequence order: the run is ONE sequence
                        for (uint32_t i = 0; i < cnt; ++i) {
                            const float x = sp[(size_t)i * kStride];
                            if (x > best) best = x;
                        }
                    } else {
                        for (uint32_t i = 0; i < cnt; ++i) {
                            const int32_t sq = row_seq[row0 + i];
                            if (sq != cur) {
                                if (best > 0.f) atomicMax(outp + (size_t)cur * ldc, __float_as_uint(best));
                                cur = sq;
                                best = 0.f;
                            }
                            const float x = sp[(size_t)i * kStride];
                            if (x > best) best = x;
                        }
                    }
                    if (best > 0.f) atomicMax(outp + (size_t)cur * ldc, __float_as_uint(best));
                }
            }
            if (p < 3) __syncthreads();
        }
        return;
    }
    if constexpr (OUT == GEMM_OUT_QKV) {
        // The tile = [one whole q or k head: 256 columns | 64 columns of v] (weights in tile order, launch_permute_qkv_rows):
        // head tile t goes to qkv columns [256 t, + 256) after RMSNorm (1 + w), RoPE and (q heads) the attention scale, the
        // v slice to columns [256 (heads + kv) + 64 t, + 64) as it is.  Arithmetic = qk_norm_rope_block / load_q_fragments
        // (embed_kernels.hip) on the bf16-rounded projection, so the attention kernels read what they used to compute.
        // Two passes of 128 rows through the bf16 stage; then wave w takes stage rows 16 w .. 16 w + 15 two at a time: lanes
        // 0-31 one row, lanes 32-63 the next; lane (l = lane & 31) owns dims 4 l + {0..3} and their rotation partners 128 + ...
        // TN = 4 (256-column tiles, the projection's rows in their natural order: partly filled rounds, M <~ 13k): a tile is
        // exactly one head - or 256 columns of v, copied as they are.
        static_assert(TN == 5 || TN == 4, "QKV epilogue: 256 (+ 64) columns per tile");
        const uint32_t tcol = n0 / (uint32_t)BN;
        const uint32_t nh = epi->heads + epi->kv_heads;
        const bool is_q = tcol < epi->heads;
        const bool is_head = TN == 5 || tcol < nh;                  // (workgroup-uniform)
        const float* const wnorm = is_q ? epi->wq : epi->wk;
        const float qs = is_q ? epi->q_scale : 1.0f;
        const float eps = epi->eps;
        const int hw = lane >> 5, l31 = lane & 31;
        f4 w1lo = *(const f4*)(wnorm + 4 * l31), w1hi = *(const f4*)(wnorm + 128 + 4 * l31);
#pragma unroll
        for (int e = 0; e < 4; ++e) { w1lo[e] = 1.0f + w1lo[e]; w1hi[e] = 1.0f + w1hi[e]; }
        constexpr int kStride = BN + 4;                             // elements (8-byte aligned rows, bank shift of 2 dwords)
        bf16_t* const stage = p8smem;                               // 128 x kStride elements = 81 KiB
        bf16_t* const Cq = (bf16_t*)Cv;
        const uint32_t vcol0 = TN == 5 ? nh * 256u + tcol * 64u : n0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // this pass's rows of the wave: positions + cos / sin rows requested before the stage is written
            uint32_t grow[8];
            f4 cs0[8], cs1[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const uint32_t sr = (uint32_t)(16 * wid + 2 * it + hw);
                grow[it] = m0 + (sr >> 6) * 128u + (uint32_t)(64 * p) + (sr & 63u);
                const uint32_t gr = grow[it] < M ? grow[it] : M - 1u;
                const float* cs = epi->cos_sin + ((size_t)(is_head ? (uint32_t)epi->pos[gr] : 0u) * 128u + (uint32_t)l31 * 4u) * 2u;
                cs0[it] = *(const f4*)cs;
                cs1[it] = *(const f4*)(cs + 4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16_t* rowp = stage + (size_t)(wm * 64 + t * 16 + l15) * kStride;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bf4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[4 * p + t][j][r];
                    *(bf4*)(rowp + wn * 16 * TN + j * 16 + 4 * lg) = o;
                }
            }
            __syncthreads();
            if (is_head)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const uint32_t sr = (uint32_t)(16 * wid + 2 * it + hw);
                const bf4 lo = *(const bf4*)(stage + (size_t)sr * kStride + 4 * l31);
                const bf4 hi = *(const bf4*)(stage + (size_t)sr * kStride + 128 + 4 * l31);
                // packed f32 arithmetic (two elements per instruction; the same operations in the same order per element
                // as the scalar form: v * inv * (1 + w), then the rotation, then the scale)
                p8_f2 vlo[2], vhi[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    vlo[h][0] = (float)lo[2 * h]; vlo[h][1] = (float)lo[2 * h + 1];
                    vhi[h][0] = (float)hi[2 * h]; vhi[h][1] = (float)hi[2 * h + 1];
                }
                p8_f2 s2 = vlo[0] * vlo[0];
                s2 = __builtin_elementwise_fma(vlo[1], vlo[1], s2);
                s2 = __builtin_elementwise_fma(vhi[0], vhi[0], s2);
                s2 = __builtin_elementwise_fma(vhi[1], vhi[1], s2);
                const float ss = half_wave_sum32(s2[0] + s2[1]);             // over the row's 32 lanes (DPP + one permlane swap)
                const float inv = rsqrtf(ss / 256.0f + eps);
                const p8_f2 inv2 = (p8_f2)(inv), qs2 = (p8_f2)(qs);
                // (cos, sin) x 4 dims: cs0 = (c0, s0, c1, s1), cs1 = (c2, s2, c3, s3)
                const p8_f2 c2[2] = {{cs0[it][0], cs0[it][2]}, {cs1[it][0], cs1[it][2]}};
                const p8_f2 n2[2] = {{cs0[it][1], cs0[it][3]}, {cs1[it][1], cs1[it][3]}};
                bf4 olo, ohi;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const p8_f2 w1l = {w1lo[2 * h], w1lo[2 * h + 1]}, w1h = {w1hi[2 * h], w1hi[2 * h + 1]};
                    const p8_f2 nlo = vlo[h] * inv2 * w1l;
                    const p8_f2 nhi = vhi[h] * inv2 * w1h;
                    const p8_f2 rl = (nlo * c2[h] - nhi * n2[h]) * qs2;       // d < 128: n cos - x[d + 128] sin
                    const p8_f2 rh = (nhi * c2[h] + nlo * n2[h]) * qs2;       // d >= 128: n cos + x[d - 128] sin
                    olo[2 * h] = (bf16_t)rl[0]; olo[2 * h + 1] = (bf16_t)rl[1];
                    ohi[2 * h] = (bf16_t)rh[0]; ohi[2 * h + 1] = (bf16_t)rh[1];
                }
                if (grow[it] < M) {
                    bf16_t* dst = Cq + (size_t)grow[it] * ldc + tcol * 256u + 4u * (uint32_t)l31;
                    *(bf4*)dst = olo;
                    *(bf4*)(dst + 128) = ohi;
                }
            }
            if constexpr (TN == 5) {
                // the v slice: 128 rows x 64 columns = 2048 8-byte pieces, 16 lanes per row (one 128-byte line)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t c = (uint32_t)tid + 512u * (uint32_t)u;
                    const uint32_t sr = c >> 4, ch = c & 15u;
                    const uint32_t row = m0 + (sr >> 6) * 128u + (uint32_t)(64 * p) + (sr & 63u);
                    if (row < M) *(bf4*)(Cq + (size_t)row * ldc + vcol0 + ch * 4u) = *(const bf4*)(stage + (size_t)sr * kStride + 256u + ch * 4u);
                }
            } else if (!is_head) {
                // a whole tile of v: 128 rows x 256 columns = 8192 8-byte pieces, a wave per row (512 bytes)
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t c = (uint32_t)tid + 512u * (uint32_t)u;
                    const uint32_t sr = c >> 6, ch = c & 63u;
                    const uint32_t row = m0 + (sr >> 6) * 128u + (uint32_t)(64 * p) + (sr & 63u);
                    if (row < M) *(bf4*)(Cq + (size_t)row * ldc + vcol0 + ch * 4u) = *(const bf4*)(stage + (size_t)sr * kStride + ch * 4u);
                }
            }
            if (p == 0) __syncthreads();
        }
        return;
    }
    if constexpr (OUT == GEMM_OUT_GEGLU4) {
        // W rows interleaved per 4 (launch_permute_geglu_rows): in a 16-column n-tile, lane group lg holds gate channels
        // c + 0..3 (lg = 0), up c + 0..3 (lg = 1), gate c + 4..7 (lg = 2), up c + 4..7 (lg = 3), c = 8 x the tile's index.
        // v_permlane16_swap(a, b) exchanges a's odd 16-lane rows with b's even rows: applied to accumulator registers (0, 2)
        // and (1, 3) it leaves lane group lg with gate AND up of channels c + 2 lg and c + 2 lg + 1 - the pairing costs two
        // instructions per tile, gelu runs on the accumulators themselves, and the stage holds the bf16 result (half the
        // columns, two passes) instead of f32 pairs (four passes): round 4, after timing builds put the old epilogue at 24 of
        // the GeGLU projection's 68 us (6.5 us of them the f32 stage).
        constexpr int ON = BN / 2;                                  // output channels of the tile
        constexpr int kStride = ON + 4;                             // elements
        constexpr int kCPR = ON / 4;                                // 8-byte chunks per row
        bf16_t* const stage = p8smem;                               // 128 x kStride elements <= 42 KiB
        typedef unsigned pu2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16_t* rowp = stage + (size_t)(wm * 64 + t * 16 + l15) * kStride;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f4 a = acc[4 * p + t][j];
                    const pu2 s02 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[0]), __float_as_uint(a[2]), false, false);
                    const pu2 s13 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[1]), __float_as_uint(a[3]), false, false);
                    p8_f2 gt, up;
                    gt[0] = __uint_as_float(s02[0]); gt[1] = __uint_as_float(s13[0]);                 // gate of channels c + 2 lg, + 1
                    up[0] = __uint_as_float(s02[1]); up[1] = __uint_as_float(s13[1]);
#if defined(P8_GEGLU4_SCALAR_GELU)   // (round 2's formula, one channel at a time: bit-identical to the f32-stage epilogue)
                    const p8_f2 res = {p8_gelu_tanh(gt[0]) * up[0], p8_gelu_tanh(gt[1]) * up[1]};
#else
                    const p8_f2 res = p8_geglu2(gt, up);
#endif
                    bf2 o;
                    o[0] = (bf16_t)res[0];
                    o[1] = (bf16_t)res[1];
                    *(bf2*)(rowp + (wn * TN + j) * 8 + 2 * lg) = o;
                }
            }
            __syncthreads();
            for (uint32_t c = (uint32_t)tid; c < 128u * (uint32_t)kCPR; c += 512u) {
                const uint32_t r = c / (uint32_t)kCPR, cc = c % (uint32_t)kCPR;
                const uint32_t row = m0 + (r >> 6) * 128u + (uint32_t)(4 * p) * 16u + (r & 63u);
                if (row < M) *(bf4*)((bf16_t*)Cv + (size_t)row * ldc + n0 / 2u + cc * 4u) = *(const bf4*)(stage + (size_t)r * kStride + cc * 4u);
            }
            if (p == 0) __syncthreads();
        }
        return;
    }
    if (OUT == GEMM_OUT_BF16 || OUT == GEMM_OUT_BF16_GELU) {
        f4 bv[TN];                                                  // this lane's 4 columns of each n-tile
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bv[j] = bias ? *(const f4*)(bias + n0 + (uint32_t)(wn * 16 * TN + j * 16 + 4 * lg)) : (f4)(0.f);
        // two passes of 128 rows (m-tiles 4p .. 4p+3 of both wave rows); LDS row = BN bf16 + 8 bytes (bank shift of
        // 2 dwords per row: conflict-free 8-byte writes)
        constexpr int kStride = BN + 4;                             // elements
        constexpr int kCPR = BN / 4;                                // 8-byte chunks per row
        bf16_t* const stage = p8smem;                               // 128 x kStride elements <= 81 KiB
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16_t* rowp = stage + (size_t)(wm * 64 + t * 16 + l15) * kStride;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bf4 o;
                    if (OUT == GEMM_OUT_BF16_GELU) {               // two values per instruction (round 4: the epilogue is VALU-bound)
                        const f4 v = acc[4 * p + t][j] + bv[j];
                        const p8_f2 g0 = p8_gelu_erf2((p8_f2){v[0], v[1]}), g1 = p8_gelu_erf2((p8_f2){v[2], v[3]});
                        o[0] = (bf16_t)g0[0]; o[1] = (bf16_t)g0[1]; o[2] = (bf16_t)g1[0]; o[3] = (bf16_t)g1[1];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[4 * p + t][j][r] + bv[j][r]);
                    }
                    *(bf4*)(rowp + wn * 16 * TN + j * 16 + 4 * lg) = o;
                }
            }
            __syncthreads();
            {
                for (uint32_t c = (uint32_t)tid; c < 128u * (uint32_t)kCPR; c += 512u) {
                    const uint32_t r = c / (uint32_t)kCPR, cc = c % (uint32_t)kCPR;
                    const uint32_t row = m0 + (r >> 6) * 128u + (uint32_t)(4 * p) * 16u + (r & 63u);
#ifdef P8_ABLATE_NOSTORE
                    if (acc[0][0][0] != 12345.678f) continue;
#endif
                    if (row < M) *(bf4*)((bf16_t*)Cv + (size_t)row * ldc + n0 + cc * 4u) = *(const bf4*)(stage + (size_t)r * kStride + cc * 4u);
                }
            }
            if (p == 0) __syncthreads();
        }
    } else {
        // GeGLU: W rows are interleaved per 64 (32 gate rows, then the same channels' 32 up rows; embedder.hip), so a
        // tile of 64 TN columns holds TN whole groups but a wave's 16 TN columns need not: the f32 accumulators go to LDS
        // and the pairing happens on the way out.  Four passes of 64 rows (m-tiles 2p, 2p+1 of both wave rows); LDS row
        // = BN f32 + 16 bytes (bank shift of 4 dwords per row: conflict-free 16-byte writes).
        constexpr int kStride = BN + 4;                             // floats
        constexpr int kGPR = BN / 8;                                // groups of 4 output channels per row
        float* const stage = (float*)p8smem;                        // 64 x kStride floats <= 66 KiB
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float* rowp = stage + (size_t)(wm * 32 + t * 16 + l15) * kStride;
#pragma unroll
                for (int j = 0; j < TN; ++j) *(f4*)(rowp + wn * 16 * TN + j * 16 + 4 * lg) = acc[2 * p + t][j];
            }
            __syncthreads();
            for (uint32_t c = (uint32_t)tid; c < 64u * (uint32_t)kGPR; c += 512u) {
                const uint32_t r = c / (uint32_t)kGPR, g = c % (uint32_t)kGPR;
                const uint32_t row = m0 + (r >> 5) * 128u + (uint32_t)(2 * p) * 16u + (r & 31u);
                const uint32_t ch = g * 4u;                          // output channel inside the tile (0 .. BN/2)
                const float* src = stage + (size_t)r * kStride + (ch >> 5) * 64u + (ch & 31u);
                const f4 gate = *(const f4*)src, up = *(const f4*)(src + 32);
                bf4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(p8_gelu_tanh(gate[e]) * up[e]);
#ifdef P8_ABLATE_NOSTORE
                if (acc[0][0][0] != 12345.678f) continue;
#endif
                if (row < M) *(bf4*)((bf16_t*)Cv + (size_t)row * ldc + n0 / 2u + ch) = o;
            }
            if (p < 3) __syncthreads();
        }
    }
}

template <int TN, int OUT>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                         void* __restrict__ Cv, uint32_t M, uint32_t N, uint32_t K,
                                                         uint32_t ldc, const float* __restrict__ bias,
                                                         const int32_t* __restrict__ row_seq, uint32_t n_valid) {
    gemm_pp_body<TN, OUT>(A, W, Cv, M, N, K, ldc, blockIdx.x, bias, row_seq, n_valid);
}

// The QKV projection with its fused epilogue: the same body, the epilogue's inputs by value in kernel-argument memory.
template <int TN>
__global__ __launch_bounds__(512, 2) void gemm_pp_qkv_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                             bf16_t* __restrict__ qkv, uint32_t M, uint32_t N, uint32_t K,
                                                             uint32_t ldc, const QkvEpilogue epi) {
    gemm_pp_body<TN, GEMM_OUT_QKV>(A, W, qkv, M, N, K, ldc, blockIdx.x, nullptr, nullptr, 0u, &epi);
}

// Wf rows in tile order: tile t (t < heads + kv) = [the 256 rows of head slot t | v rows 64 t .. 64 t + 63]; one thread per 16 bytes.
__global__ __launch_bounds__(256) void permute_qkv_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                               uint32_t nh, uint32_t K) {
    const uint32_t chunk = blockIdx.x * 256u + threadIdx.x, cpr = K / 8u;
    if (chunk >= nh * 320u * cpr) return;
    const uint32_t drow = chunk / cpr, c = chunk % cpr;
    const uint32_t t = drow / 320u, r = drow % 320u;
    const uint32_t srow = r < 256u ? t * 256u + r : nh * 256u + t * 64u + (r - 256u);
    *(bf8*)(dst + (size_t)drow * K + (size_t)c * 8u) = *(const bf8*)(src + (size_t)srow * K + (size_t)c * 8u);
}

// Two problems that share A, M and K (the two column ranges launch_gemm_bf16 cuts a GEMM into: whole rounds of one
// tile width + the rest with another) in ONE launch: workgroups [0, n_first) run the first, the rest the second, so
// the second part's workgroups start as CUs come free instead of after a kernel boundary.
template <int TNA, int TNB, int OUT>
__global__ __launch_bounds__(512, 2) void gemm_pp_dual_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wa,
                                                              void* __restrict__ Ca, uint32_t Na,
                                                              const bf16_t* __restrict__ Wb, void* __restrict__ Cb,
                                                              uint32_t Nb, uint32_t M, uint32_t K, uint32_t ldc,
                                                              uint32_t n_first) {
    // (Round 4 tried the two problems' workgroups interleaved in groups of 8, so that each round holds wide and narrow tiles
    // and the CUs' load / store bursts drift apart: 68.1 -> 72.0 us for the GeGLU projection - the four column tiles of a
    // row block no longer run side by side on one XCD and its A panel is fetched into more L2s.  One after the other it stays.
    // Also tried: 256 persistent workgroups, each walking its wide tile and then the narrow tile of the same row block -
    // 57.7-58.6 us against 57.4-57.7 for this kernel: a second round of workgroup launches costs nothing measurable, so the
    // only thing a persistent form could still win is the next tile's ~2 us prologue under this tile's epilogue.)
    if (blockIdx.x < n_first) gemm_pp_body<TNA, OUT>(A, Wa, Ca, M, Na, K, ldc, blockIdx.x);
    else gemm_pp_body<TNB, OUT>(A, Wb, Cb, M, Nb, K, ldc, blockIdx.x - n_first);
}

template <int TNA, int TNB, int OUT>
hipError_t launch_p8_dual(const bf16_t* A, const bf16_t* Wa, void* Ca, uint32_t Na, const bf16_t* Wb, void* Cb, uint32_t Nb,
                          uint32_t M, uint32_t K, uint32_t ldc, hipStream_t st) {
    constexpr int BNA = 64 * TNA, BNB = 64 * TNB, BNX = BNA > BNB ? BNA : BNB;
    const uint32_t mt = (M + kP8M - 1) / kP8M;
    const uint32_t n_first = (Na / BNA) * mt, n_second = (Nb / BNB) * mt;
    const size_t lds = (size_t)2 * (kP8M + BNX) * 64 * sizeof(bf16_t);
    auto kern = gemm_pp_dual_kernel<TNA, TNB, OUT>;
    static std::atomic<uint64_t> attr_devices{0};   // per instantiation, per device
    {
        const hipError_t e = set_max_dynamic_lds((const void*)kern, lds, attr_devices);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_first + n_second), dim3(512), lds, st, A, Wa, Ca, Na, Wb, Cb, Nb, M, K, ldc, n_first);
    return hipGetLastError();
}

template <int TN, int OUT>
hipError_t launch_p8(const bf16_t* A, const bf16_t* W, void* C, uint32_t M, uint32_t N, uint32_t K, uint32_t ldc,
                     const float* bias, hipStream_t st, const int32_t* row_seq = nullptr, uint32_t n_valid = 0) {
    constexpr int BN = 64 * TN;
    const dim3 grid((N / BN) * ((M + kP8M - 1) / kP8M));
    const size_t lds = (size_t)2 * (kP8M + BN) * 64 * sizeof(bf16_t);
    auto kern = gemm_pp_kernel<TN, OUT>;
    static std::atomic<uint64_t> attr_devices{0};   // per instantiation, per device
    {
        const hipError_t e = set_max_dynamic_lds((const void*)kern, lds, attr_devices);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, A, W, C, M, N, K, ldc, bias, row_seq, n_valid);
    return hipGetLastError();
}

}  // namespace

// gate / up rows: interleave per 32 (src) -> per 4 (dst); one thread per 16 bytes
__global__ __launch_bounds__(256) void permute_geglu_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, uint32_t N, uint32_t K) {
    const uint32_t chunk = blockIdx.x * 256u + threadIdx.x, cpr = K / 8u;
    if (chunk >= N * cpr) return;
    const uint32_t d = chunk / cpr, c = chunk % cpr;
    const uint32_t jt = d / 16u, q = (d % 16u) / 4u, r = d % 4u;
    const uint32_t ch = 8u * jt + 4u * (q >> 1) + r, up = q & 1u;
    const uint32_t srow = 64u * (ch / 32u) + 32u * up + ch % 32u;
    *(bf8*)(dst + (size_t)d * K + (size_t)c * 8u) = *(const bf8*)(src + (size_t)srow * K + (size_t)c * 8u);
}
hipError_t launch_permute_geglu_rows(const bf16_t* src, bf16_t* dst, uint32_t N, uint32_t K, hipStream_t st) {
    if (!src || !dst || N % 64u || K % 8u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(permute_geglu_rows_kernel, dim3((N * (K / 8u) + 255u) / 256u), dim3(256), 0, st, src, dst, N, K);
    return hipGetLastError();
}

hipError_t launch_permute_qkv_rows(const bf16_t* wqkv, bf16_t* wf, uint32_t heads, uint32_t kv_heads, uint32_t K, hipStream_t st) {
    if (!wqkv || !wf || heads != 3u * kv_heads || K % 8u) return hipErrorInvalidValue;
    const uint32_t nh = heads + kv_heads;
    hipLaunchKernelGGL(permute_qkv_rows_kernel, dim3((nh * 320u * (K / 8u) + 255u) / 256u), dim3(256), 0, st, wqkv, wf, nh, K);
    return hipGetLastError();
}

hipError_t launch_gemm_qkv_rope(const bf16_t* A, const bf16_t* W, bf16_t* qkv, uint32_t M, uint32_t K, int tn, const QkvEpilogue& epi,
                                hipStream_t st) {
    if (M == 0) return hipSuccess;
    const uint32_t nh = epi.heads + epi.kv_heads, N = (epi.heads + 2u * epi.kv_heads) * 256u;
    if ((tn != 4 && tn != 5) || (tn == 5 && epi.heads != 3u * epi.kv_heads) || K % 64u || K < 64u || (uint64_t)M * K >= (1ull << 31) ||
        (uint64_t)N * K >= (1ull << 31) || !epi.pos || !epi.wq || !epi.wk || !epi.cos_sin)
        return hipErrorInvalidValue;
    const uint32_t mt = (M + kP8M - 1) / kP8M;
    if (tn == 5) {
        const size_t lds = (size_t)2 * (kP8M + 320) * 64 * sizeof(bf16_t);
        static std::atomic<uint64_t> attr_devices{0};
        const hipError_t e = set_max_dynamic_lds((const void*)gemm_pp_qkv_kernel<5>, lds, attr_devices);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gemm_pp_qkv_kernel<5>, dim3(nh * mt), dim3(512), lds, st, A, W, qkv, M, N, K, N, epi);
    } else {
        const size_t lds = (size_t)2 * (kP8M + 256) * 64 * sizeof(bf16_t);
        static std::atomic<uint64_t> attr_devices{0};
        const hipError_t e = set_max_dynamic_lds((const void*)gemm_pp_qkv_kernel<4>, lds, attr_devices);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gemm_pp_qkv_kernel<4>, dim3((N / 256u) * mt), dim3(512), lds, st, A, W, qkv, M, N, K, N, epi);
    }
    return hipGetLastError();
}

// tn: n-tiles per wave (tile width 64 tn: 192 / 256 / 320); N % (64 tn) == 0, K % 64 == 0, M * K < 2^31 elements.
hipError_t launch_gemm_p8(const bf16_t* A, const bf16_t* W, void* C, uint32_t M, uint32_t N, uint32_t K, uint32_t ldc,
                          GemmOut out, int tn, hipStream_t st, const float* bias) {
    if (M == 0) return hipSuccess;
    if (K % 64u || K < 64u || tn < 3 || tn > 5 || N % (64u * (uint32_t)tn) || (uint64_t)M * K >= (1ull << 31) ||
        (uint64_t)N * K >= (1ull << 31))
        return hipErrorInvalidValue;
#define P8_CASE(TNV)                                                                                      \
    case TNV:                                                                                             \
        if (out == GEMM_OUT_BF16) return launch_p8<TNV, GEMM_OUT_BF16>(A, W, C, M, N, K, ldc, bias, st);  \
        if (out == GEMM_OUT_F32) return launch_p8<TNV, GEMM_OUT_F32>(A, W, C, M, N, K, ldc, bias, st);    \
        if (out == GEMM_OUT_BF16_GELU) return launch_p8<TNV, GEMM_OUT_BF16_GELU>(A, W, C, M, N, K, ldc, bias, st); \
        if (bias) return hipErrorInvalidValue;                                                            \
        if (out == GEMM_OUT_GEGLU4) return launch_p8<TNV, GEMM_OUT_GEGLU4>(A, W, C, M, N, K, ldc, nullptr, st); \
        return launch_p8<TNV, GEMM_OUT_GEGLU>(A, W, C, M, N, K, ldc, nullptr, st);
    switch (tn) {
        P8_CASE(3)
        P8_CASE(4)
        P8_CASE(5)
        default: break;
    }
#undef P8_CASE
    return hipErrorInvalidValue;
}

// out_bits[row_seq[m]][v] = max(out_bits, bits of max(0, bf16(A W^T + bias)[m][v])) for v < n_valid: per-sequence column
// maxima of a GEMM whose result is never stored (the SPLADE decoder + pooling).  out_bits [sequences, ldc] u32, zeroed by
// the caller; N % 192 == 0.
hipError_t launch_gemm_rowmax(const bf16_t* A, const bf16_t* W, const float* bias, uint32_t* out_bits, uint32_t M, uint32_t N,
                              uint32_t K, uint32_t ldc, const int32_t* row_seq, uint32_t n_valid, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (K % 64u || K < 64u || N % 192u || !row_seq || (uint64_t)M * K >= (1ull << 31) || (uint64_t)N * K >= (1ull << 31))
        return hipErrorInvalidValue;
    return launch_p8<3, GEMM_OUT_ROWMAX>(A, W, out_bits, M, N, K, ldc, bias, st, row_seq, n_valid);
}

// Two column ranges of one GEMM (same A, M, K, ldc) with different tile widths in one launch; tn_a != tn_b, both in
// {3, 4, 5}; out = BF16 or GEGLU.  hipErrorNotSupported: the caller launches the two parts separately.
hipError_t launch_gemm_p8_dual(const bf16_t* A, const bf16_t* Wa, void* Ca, uint32_t Na, int tn_a, const bf16_t* Wb,
                               void* Cb, uint32_t Nb, int tn_b, uint32_t M, uint32_t K, uint32_t ldc, GemmOut out,
                               hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (tn_a < tn_b) { std::swap(Wa, Wb); std::swap(Ca, Cb); std::swap(Na, Nb); std::swap(tn_a, tn_b); }
    if (out == GEMM_OUT_F32 || tn_a == tn_b || tn_b < 3 || tn_a > 5) return hipErrorNotSupported;
    if (K % 64u || K < 64u || Na % (64u * (uint32_t)tn_a) || Nb % (64u * (uint32_t)tn_b) || (uint64_t)M * K >= (1ull << 31) ||
        (uint64_t)Na * K >= (1ull << 31) || (uint64_t)Nb * K >= (1ull << 31))
        return hipErrorInvalidValue;
    if ((Na / (64u * (uint32_t)tn_a)) * ((M + kP8M - 1) / kP8M) % 8u) return hipErrorNotSupported;   // second part must start on XCD 0
#define P8_DUAL(TA, TB)                                                                                              \
    if (tn_a == TA && tn_b == TB)                                                                                    \
        return out == GEMM_OUT_BF16 ? launch_p8_dual<TA, TB, GEMM_OUT_BF16>(A, Wa, Ca, Na, Wb, Cb, Nb, M, K, ldc, st) \
               : out == GEMM_OUT_GEGLU4 ? launch_p8_dual<TA, TB, GEMM_OUT_GEGLU4>(A, Wa, Ca, Na, Wb, Cb, Nb, M, K, ldc, st) \
                                        : launch_p8_dual<TA, TB, GEMM_OUT_GEGLU>(A, Wa, Ca, Na, Wb, Cb, Nb, M, K, ldc, st);
    P8_DUAL(5, 4)
    P8_DUAL(5, 3)
    P8_DUAL(4, 3)
#undef P8_DUAL
    return hipErrorNotSupported;
}

}  // namespace cqs
