// gemm_rowfuse.hip — a branch's output projection FUSED with the residual add and the two RMSNorms around it:
//
//     y      = A W^T                                (o_proj: A = attention rows; down: A = GeGLU rows)
//     x     += rmsnorm(bf16(y)) (1 + w_post)        (the f32 residual stream, in place)
//     xn     = bf16(rmsnorm(x) (1 + w_next))        (FINAL: out = f32 of the same, the model's final norm)
//
// i.e. `gemm_pp_kernel<.., GEMM_OUT_BF16>` + `add_norm_kernel` of the unfused chain in ONE launch, bit for bit the same
// numbers (same MFMA, same k order -> the same bf16 y; the row phase is add_norm_kernel's arithmetic on y rows that
// never left the CU).  Why: VERDICT r02 / profiles/r02_embed_rocprofv3.txt - the two add_norm launches of a layer move
// 302 of the layer's 731 MB (41 %) and take 46 of its 254 us at the HBM roof, and every GEMM launch idles the matrix
// pipe ~60 % around its main loop.  Fused, the y round trip (a 25 MB write + a 25 MB read per projection) and two
// launches per layer disappear; what remains of the add_norm traffic (x in / x out / xn out) is the residual stream.
//
// A row norm needs the whole row, so a workgroup must own whole rows: tile = 64 token rows x ALL 768 columns, 256
// workgroups at 16 384 tokens = one per CU.  8 waves; wave w owns columns [96 w, 96 w + 96) = 6 n-tiles x 4 m-tiles of
// 16 x 16 (96 accumulator registers).  Per 32-deep k-step a workgroup needs 64 x 32 of A (4 KB) and ALL of W's 768 x 32
// (48 KB): the W panel (1.18 / 1.77 MB) stays in every XCD's L2 and streams through every CU once per launch - that
// stream, ~52 KB per 768 MFMA clocks = 68 B/clk/CU against the ~56 B/clk an XCD's L2 delivers, bounds the main loop
// (~11 us at K = 768, ~16 us at K = 1152; the 256-row tiles of gemm_kernels.hip need a quarter of that per flop).
//   * operands by LDS-DMA (global_load_lds_dwordx4 from inline asm, counted vmcnt) into a ring of THREE stages (156 KB):
//     the DMA runs two k-steps ahead of the MFMAs;
//   * a wave DMAs and reads its OWN 96 W rows: only the 4 KB A tile is shared, so one barrier per k-step;
//   * 64-byte LDS rows (32 bf16), 16-byte chunk c of row R stored at chunk c ^ f[(R >> 2) & 3], f = {0, 2, 3, 1}:
//     the 16 lanes a ds_read_b128 serves per cycle ({0-3, 12-15, 20-27}, ...) then hit 16 distinct 16-byte bank slots;
//     the swizzle is applied to the DMA's SOURCE address (its LDS side is linear);
//   * epilogue: accumulators -> bf16 -> LDS [64][776] (reusing the ring), barrier, then wave w takes rows w, w + 8, ...
//     exactly like add_norm_kernel (lane owns 4 consecutive floats of each 256-chunk), x loads issued before the barrier.
#include "embed_kernels.h"
#include "launch_util.h"

#include <type_traits>

namespace cqs {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kRfRows = 64;                              // token rows per workgroup
constexpr int kRfH = 768;                                // output columns = hidden
constexpr int kRfSlot = 16 * 128;                        // one W slot: 16 rows x 64 k (128-byte rows) = 2 KB
constexpr int kRfWRing = 8 * 6 * kRfSlot;                // 8 waves x 6 private slots = 96 KB
constexpr int kRfATile = kRfRows * 128;                  // the shared A tile of one k-step: 8 KB, double-buffered
constexpr int kRfLds = kRfWRing + 2 * kRfATile;          // 112 KB
constexpr int kRfLdy = kRfH + 8;                         // epilogue row stride (elements): 64 x 776 x 2 B = 97 KB <= kRfLds

__device__ __forceinline__ float rf_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

#define RF_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// Time structure of the main loop (64-deep k-steps; every LDS row is one full 128-byte line of its matrix - 64-byte rows
// made every DMA instruction a 16-segment gather, ~44 clocks of address processing each: 3 000 clocks per k-step):
//   * W: wave w streams its OWN 96 rows through 6 private slots (slot j = n-tile j: 16 rows x 64 k, two DMA
//     instructions of 8 rows x 128 B).  Step (kt, j): wait for slot (kt, j), read its two fragments, refill the slot with
//     (kt + 1, j), 8 MFMAs.  Nobody else touches a wave's slots: no barrier on the W stream, 5 slots (10 KB) always in flight.
//   * A: the 64 x 64 tile of a k-step is shared: one DMA instruction per wave into a double buffer at the top of the
//     step before (right behind that step's barrier), one barrier per k-step.
//   * counted waits, constant through the whole loop because the tail keeps issuing (harmless) refills of the last
//     k-step: DMA completes in issue order, and between the DMA of slot (kt, j) and its use this wave issues slots
//     (kt, j+1..5), A (kt + 1) and slots (kt + 1, 0..j-1) = 2 (5 - j) + 1 + 2 j = 11 instructions -> vmcnt(11); behind A (kt)
//     come the 12 refill instructions of step kt - 1 -> vmcnt(12) at the top of step kt.
template <int FINAL>
__global__ __launch_bounds__(512, 2) void gemm_rowfuse_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              float* __restrict__ x, const float* __restrict__ w_post,
                                                              const float* __restrict__ w_next, float eps,
                                                              bf16_t* __restrict__ xn, float* __restrict__ out,
                                                              uint32_t M, uint32_t K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const uint32_t m0 = blockIdx.x * (uint32_t)kRfRows;
    const uint32_t rows_left = M - m0;                   // >= 1
    const uint32_t nk = K / 64u;

    // ---- DMA sources.  One instruction = 8 LDS rows x 128 B: lane -> row (lane >> 3), physical chunk (lane & 7) ->
    // logical chunk (lane & 7) ^ ((R >> 1) & 7), R = the row's index inside its 16-row group (embed_kernels.hip swz()).
    const uint32_t r8 = (uint32_t)lane >> 3;
    const uint32_t pc = (uint32_t)lane & 7u;
    // W slot j, half hf (rows 8 hf + r8 of n-tile j): row 96 wid + 16 j + 8 hf + r8; the (j, hf, kt) parts are scalar
    uint32_t voffW[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const uint32_t R = (uint32_t)(8 * hf) + r8;
#if defined(CQS_RF_ABLATE_SAMEW)   // timing experiment: every wave streams wave 0's rows (1 / 8 of the distinct lines per CU)
        voffW[hf] = (R * K + ((pc ^ ((R >> 1) & 7u)) * 8u)) * 2u;
#else
        voffW[hf] = (((uint32_t)(96 * wid) + R) * K + ((pc ^ ((R >> 1) & 7u)) * 8u)) * 2u;
#endif
    }
    // A: wave w covers tile rows 8 w + r8 (R = that row & 15)
    uint32_t arow = (uint32_t)(8 * wid) + r8;
    const uint32_t aR = arow & 15u;
    if (arow >= rows_left) arow = rows_left - 1u;        // rows past M: any real row (their outputs are never stored)
    const uint32_t voffA = (arow * K + ((pc ^ ((aR >> 1) & 7u)) * 8u)) * 2u;
    const char* const gA = (const char*)(A + (size_t)m0 * K);
    const char* const gW = (const char*)W;
    const size_t rowK16 = (size_t)K * 32u;               // bytes of 16 rows
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)rf_smem;
    auto dma = [&](const char* sbase, uint32_t voff, uint32_t lds_byte) {
        const uint32_t m0v = lds0 + lds_byte;
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(m0v), "v"(voff), "s"(sbase) : "memory");
    };
    const uint32_t wring = (uint32_t)(wid * 6 * kRfSlot);
    auto fill_w = [&](uint32_t kt, int j) {               // slot j <- k-step kt (two instructions)
        const char* src = gW + (size_t)kt * 128u + (size_t)j * rowK16;
        dma(src, voffW[0], wring + (uint32_t)(j * kRfSlot));
        dma(src, voffW[1], wring + (uint32_t)(j * kRfSlot + 1024));
    };
    auto fill_a = [&](uint32_t kt, uint32_t buf) {        // this wave's 8 rows of A tile kt (one instruction) into buffer buf
        dma(gA + (size_t)kt * 128u, voffA, (uint32_t)kRfWRing + buf * (uint32_t)kRfATile + (uint32_t)(wid * 1024));
    };

    // ---- fragment read offsets: row l15 of a 16-row group, k-sub s: logical chunk 4 s + lg
    const uint32_t swr = ((uint32_t)l15 >> 1) & 7u;
    uint32_t foff[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) foff[sb] = (uint32_t)l15 * 128u + ((((uint32_t)(4 * sb + lg)) ^ swr) * 16u);

    f4 acc[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[j][t] = (f4)(0.f);

    // x rows of the epilogue (wave w: rows w, w + 8, ...; lane owns floats [256 c + 4 lane, + 4)): requested before the LAST
    // k-step, so that their HBM latency (the main loop runs out of L2: HBM is idle) hides under it
    f4 xv[8][3];
    auto load_x = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t r = (uint32_t)(wid + 8 * i);
            const uint32_t row = m0 + (r < rows_left ? r : rows_left - 1u);
#pragma unroll
            for (int c = 0; c < 3; ++c) xv[i][c] = *(const f4*)(x + (size_t)row * kRfH + c * 256 + lane * 4);
        }
    };
    const uint32_t klast = nk - 1u;
    // One k-step.  Fragments of slot j + 1 are read while slot j multiplies (the LDS latency hides behind 8 MFMAs); between
    // the DMA of slot (kt, j + 1) and that read this wave issued 2 (4 - j) + 1 + 2 j = 9 instructions -> vmcnt(9); slot (kt, 0)
    // is read at the top behind 10 + 1 = 11.
    // XTRA: vector-memory instructions of the compiler's own (the 24 x loads before the last step) that sit in the same
    // in-order queue between the DMAs being waited for and the ones issued since
    auto kstep = [&](uint32_t kt, auto xtra_c) {
        constexpr int XTRA = decltype(xtra_c)::value;
        RF_WAIT_VM(12 + XTRA);                            // this wave's part of A (kt) has landed (behind it: 12 W instructions)
        asm volatile("s_barrier" ::: "memory");          // ... and everybody's; everybody is done reading A (kt - 1)
        const uint32_t kn = kt < klast ? kt + 1u : klast; // (the last step re-requests itself: keeps the wait counts constant)
        fill_a(kn, (kt + 1u) & 1u);                       // into the buffer A (kt - 1) lived in
        const unsigned char* sa = rf_smem + kRfWRing + (kt & 1u) * (uint32_t)kRfATile;
        const unsigned char* sw = rf_smem + wring;
        bf8 af[4][2], wf[2], wn[2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) af[t][sb] = *(const bf8*)(sa + (uint32_t)(t * 2048) + foff[sb]);
        RF_WAIT_VM(11 + XTRA);                            // slot (kt, 0)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) wf[sb] = *(const bf8*)(sw + foff[sb]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (j < 5) {
                RF_WAIT_VM(9 + XTRA);                     // slot (kt, j + 1)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) wn[sb] = *(const bf8*)(sw + (uint32_t)((j + 1) * kRfSlot) + foff[sb]);
            }
#if defined(CQS_RF_ABLATE_HALFDMA)   // timing experiments (wrong results): half the W refills / no MFMAs
            if ((j & 1) == 0)
#endif
            fill_w(kn, j);                                // slot j is in registers: refill it
#if defined(CQS_RF_ABLATE_NOMFMA)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[j][t][0] += (float)wf[0][t] * (float)af[t][1][0];
#else
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[sb], af[t][sb], acc[j][t], 0, 0, 0);
#endif
            if (j < 5) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wf[0] = wn[0]; wf[1] = wn[1];
            }
        }
    };
    fill_a(0u, 0u);
#pragma unroll
    for (int j = 0; j < 6; ++j) fill_w(0u, j);
    for (uint32_t kt = 0; kt < klast; ++kt) kstep(kt, std::integral_constant<int, 0>{});
    load_x();
    kstep(klast, std::integral_constant<int, 24>{});
    RF_WAIT_VM(0);                                        // the tail's surplus refills are done before LDS is reused

    // ---- epilogue ----
    __syncthreads();                                      // every wave is out of the ring
    bf16_t* const sY = (bf16_t*)rf_smem;                  // [64][kRfLdy]
    // acc[j][t][r] = y[token 16 t + l15][column 96 wid + 16 j + 4 lg + r]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            bf4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[j][t][r];
            *(bf4*)(sY + (size_t)(16 * t + l15) * kRfLdy + 96 * wid + 16 * j + 4 * lg) = o;
        }
    __syncthreads();
    f4 wpo[3], wne[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        wpo[c] = *(const f4*)(w_post + c * 256 + lane * 4);
        wne[c] = *(const f4*)(w_next + c * 256 + lane * 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r = (uint32_t)(wid + 8 * i);
        if (r >= rows_left) continue;                     // wave-uniform (a guard, not a break: the loop must unroll)
        const size_t row = (size_t)(m0 + r);
        f4 yv[3];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bf4 yb = *(const bf4*)(sY + (size_t)r * kRfLdy + c * 256 + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { yv[c][e] = (float)yb[e]; ss += yv[c][e] * yv[c][e]; }
        }
        const float invy = rsqrtf(rf_wave_sum(ss) / (float)kRfH + eps);
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[i][c][e] += yv[c][e] * invy * (1.0f + wpo[c][e]);
                sx += xv[i][c][e] * xv[i][c][e];
            }
            *(f4*)(x + row * kRfH + c * 256 + lane * 4) = xv[i][c];
        }
        const float invx = rsqrtf(rf_wave_sum(sx) / (float)kRfH + eps);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (FINAL) {
                f4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = xv[i][c][e] * invx * (1.0f + wne[c][e]);
                *(f4*)(out + row * kRfH + c * 256 + lane * 4) = o;
            } else {
                bf4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(xv[i][c][e] * invx * (1.0f + wne[c][e]));
                *(bf4*)(xn + row * kRfH + c * 256 + lane * 4) = o;
            }
        }
    }
}


// =====================================================================================================================
// Second generation (round 4): a PAIR of workgroups owns 128 token rows; each computes one 384-column HALF of all 768.
//
// Why: the 64-row tile above streams the whole W panel (96 KB per 64-deep k-step) through every CU, and that stream -
// not the matrix cores, not HBM - bounds its main loop (§3.5b of DESIGN.md: ~30 B/clk per CU whatever instruction
// carries the bytes).  A workgroup that owns 128 rows x 384 columns needs W's half panel (48 KB) + a 128-row A tile
// (16 KB) per k-step for the SAME 12.6 MFLOP: 64 KB instead of 104 KB.  The price: a row's sum of squares is now split
// over two workgroups, twice per launch (y rows, then the new x rows) - exchanged as 8-byte {tag, value} granules
// (one sc1 store per granule, polled with sc1 loads: cdna_hip_programming.md Guideline 16, form R2; tag = the launch's
// sequence number, so nothing has to be zeroed between launches; every spin is bounded and a timeout raises *err).
// Partners are workgroups b and b ^ 8: under the round-robin placement they share an XCD, so the pair's A rows are
// read from HBM once (speed only; nothing depends on placement).  Partners sit within 16 consecutive workgroups, so at
// most 8 workgroups of a launch can be waiting for a partner that is not resident yet.
//   * 8 waves; wave w owns columns [48 w, 48 w + 48) of its half = 3 n-tiles, all 128 rows = 8 m-tiles: 96 accumulators;
//   * W: the wave's own 48 rows through 6 private slots = 3 n-tiles x 2 k-steps (slot (kt & 1, j) is refilled with
//     k-step kt + 2 as soon as its fragments are in registers): ~10 KB per wave always in flight, no barrier;
//   * A: the shared 128 x 64 tile of a k-step, THREE buffers, requested two k-steps ahead, one barrier per k-step;
//   * every counted wait is vmcnt(14): between a DMA and its use the wave issues exactly 14 more (see kstep);
//   * row sums half by half, left + right: the order add_norm_kernel uses too (embed_kernels.hip) -> same bits as the
//     two-launch chain.
constexpr int kR2Rows = 128;                              // token rows per workgroup pair
constexpr int kR2Half = 384;                              // columns per workgroup
constexpr int kR2Slot = 16 * 128;                         // one W slot: 16 rows x 64 k = 2 KB
constexpr int kR2WRing = 8 * 6 * kR2Slot;                 // 8 waves x (3 n-tiles x 2 k-steps) = 96 KB
constexpr int kR2ATile = kR2Rows * 128;                   // 16 KB
constexpr int kR2Lds = kR2WRing + 3 * kR2ATile;           // 144 KB
constexpr int kR2Ldy = kR2Half + 8;                       // epilogue row stride (elements): 128 x 392 x 2 B = 98 KB
constexpr uint32_t kR2SpinMax = 1u << 21;

typedef __attribute__((address_space(1))) unsigned long long r2_gu64;
typedef __attribute__((address_space(1))) unsigned r2_gu32;

// dst = the weights [N][K] in the order gemm_rowfuse2_kernel streams them: block (n / 16, k / 64) of 2 KB = 16 rows x 64 k,
// row r at byte r * 128, its 16-byte chunk c stored at chunk c ^ ((r >> 1) & 7) (the LDS swizzle, applied once here).
__global__ __launch_bounds__(256) void pack_rowfuse_w_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, uint32_t N, uint32_t K) {
    const uint32_t nk = K / 64u;
    const uint32_t chunk = blockIdx.x * 256u + threadIdx.x;           // one 16-byte chunk per thread
    if (chunk >= N * (K / 8u)) return;
    const uint32_t n = chunk / (K / 8u), kc = chunk % (K / 8u);       // row, chunk of 8 k
    const uint32_t kt = kc / 8u, c = kc % 8u, r = n % 16u, jt = n / 16u;
    const bf8 v = *(const bf8*)(src + (size_t)n * K + (size_t)kc * 8u);
    *(bf8*)(dst + ((size_t)(jt * nk + kt) * 16u + r) * 64u + ((c ^ ((r >> 1) & 7u)) * 8u)) = v;
}

template <int FINAL>
__global__ __launch_bounds__(512, 2) void gemm_rowfuse2_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W /*packed: pack_rowfuse_w_kernel*/,
                                                               float* __restrict__ x, const float* __restrict__ w_post,
                                                               const float* __restrict__ w_next, float eps,
                                                               bf16_t* __restrict__ xn, float* __restrict__ out,
                                                               uint32_t M, uint32_t K, unsigned long long* xch /*[2][2][Mpad] granules*/,
                                                               uint32_t Mpad, uint32_t tag, unsigned* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    // workgroups 16 g + i and 16 g + 8 + i (i < 8) form pair 8 g + i; the half is bit 3 of the index
    const uint32_t bid = blockIdx.x;
    const uint32_t half = (bid >> 3) & 1u;
    const uint32_t pair = (bid >> 4) * 8u + (bid & 7u);
    const uint32_t m0 = pair * (uint32_t)kR2Rows;
    if (m0 >= M) return;                                  // (an odd tail group: both partners leave together)
    const uint32_t rows_left = M - m0;                    // >= 1
    const uint32_t nk = K / 64u;
    const uint32_t n0 = half * (uint32_t)kR2Half;

    const uint32_t r8 = (uint32_t)lane >> 3;
    const uint32_t pc = (uint32_t)lane & 7u;
    // W comes PACKED (pack_rowfuse_w_kernel): the 2 KB LDS image of (n-tile, k-step) - 16 rows x 128 B, swizzle applied - is
    // one contiguous 2 KB block at ((n-tile * nk + k-step) * 2 KB), so a DMA instruction reads 1 KB in one piece instead
    // of eight 128-byte segments of eight rows (the texture addresser works per segment).
    const uint32_t voffW = (uint32_t)lane * 16u;
    // A: instruction i of 16 covers tile rows 8 i .. 8 i + 7; wave w issues i = w and i = w + 8
    uint32_t voffA[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t arow = (uint32_t)(8 * wid + 64 * u) + r8;
        const uint32_t aR = arow & 15u;
        if (arow >= rows_left) arow = rows_left - 1u;     // rows past M: any real row (their outputs are never stored)
        voffA[u] = (arow * K + ((pc ^ ((aR >> 1) & 7u)) * 8u)) * 2u;
    }
    const char* const gA = (const char*)(A + (size_t)m0 * K);
    const char* const gW = (const char*)W;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)rf_smem;
    auto dma = [&](const char* sbase, uint32_t voff, uint32_t lds_byte) {
        const uint32_t m0v = lds0 + lds_byte;
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(m0v), "v"(voff), "s"(sbase) : "memory");   // (hipcc rejects an "m0" clobber: "reserved register"; M0 is set and consumed inside this one statement)
    };
    const uint32_t wring = (uint32_t)(wid * 6 * kR2Slot);
    const uint32_t ntile0 = half * 24u + (uint32_t)(3 * wid);   // first of this wave's three 16-column tiles
    // (Timing experiment, round 4: workgroups starting their k loop at different k-steps, so that an XCD's 32 workgroups do
    // not all ask its L2 for the same W lines at the same time - main loop 17.7 vs 18.1 us: no effect, removed.)
    auto fill_w = [&](uint32_t kt, uint32_t par, int j) {  // slot (par, j) <- k-step kt
        const char* src = gW + ((ntile0 + (uint32_t)j) * nk + kt) * (uint32_t)kR2Slot;   // (32-bit: a 64-bit multiply would leave the scalar unit)
        const uint32_t lb = wring + (par * 3u + (uint32_t)j) * (uint32_t)kR2Slot;
        dma(src, voffW, lb);
        dma(src + 1024, voffW, lb + 1024u);
    };
    auto fill_a = [&](uint32_t kt, uint32_t buf) {         // this wave's two pieces of A tile kt into buffer buf
        const uint32_t lb = (uint32_t)kR2WRing + buf * (uint32_t)kR2ATile;
        dma(gA + (size_t)kt * 128u, voffA[0], lb + (uint32_t)(wid * 1024));
        dma(gA + (size_t)kt * 128u, voffA[1], lb + (uint32_t)(8192 + wid * 1024));
    };
    const uint32_t swr = ((uint32_t)l15 >> 1) & 7u;
    uint32_t foff[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) foff[sb] = (uint32_t)l15 * 128u + ((((uint32_t)(4 * sb + lg)) ^ swr) * 16u);

    f4 acc[3][8];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[j][t] = (f4)(0.f);

    const uint32_t klast = nk - 1u;
    // (Tried in round 4 and dropped: pulling the epilogue's x rows into L2 from inside the main loop - each wave touching
    // the 192 lines of its 16 x-row halves by DMA loads into a dump slot, 1-8 k-steps before the end.  The launch got
    // SLOWER: 41.5 us without, 44.6 with two thirds of the lines, 46.3-47.4 with all of them at any distance - the main
    // loop is not indifferent to HBM traffic beside it, and 6 MB of x per XCD do not survive in a 4 MB L2.)
    // One k-step.  DMA issue order of a wave, step after step: A (kt + 2) x 2, then W (kt + 2, j) x 2 behind the fragment
    // reads of slot (kt, j), j = 0, 1, 2 - eight instructions per step.  They complete in issue order, so:
    //   A (kt), issued at the top of step kt - 2: behind it W (kt, 0..2) = 6 and step kt - 1's 8          -> vmcnt(14)
    //   W (kt, j), issued in step kt - 2: behind it W (kt, j+1..2) = 2 (2 - j), step kt - 1's 8, and this step's
    //   A (kt + 2) + W (kt + 2, 0..j-1) = 2 + 2 j                                                         -> vmcnt(14)
    // The last two steps keep issuing (harmless) requests of the last k-step so that the counts never change.
    auto kstep = [&](uint32_t kt, uint32_t abuf /*kt % 3*/) {
        RF_WAIT_VM(14);                                   // this wave's pieces of A (kt)
        asm volatile("s_barrier" ::: "memory");          // ... and everybody's; everybody is done reading A (kt - 1)
        const uint32_t kn = kt + 2u <= klast ? kt + 2u : klast;
        fill_a(kn, abuf == 0u ? 2u : abuf - 1u);          // buffer (kt + 2) % 3 = (kt - 1) % 3
        const unsigned char* sa = rf_smem + kR2WRing + abuf * (uint32_t)kR2ATile;
        const uint32_t par = kt & 1u;
        const unsigned char* sw = rf_smem + wring + par * (uint32_t)(3 * kR2Slot);
        bf8 af[8][2], wf[2], wn[2];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) af[t][sb] = *(const bf8*)(sa + (uint32_t)(t * 2048) + foff[sb]);
        RF_WAIT_VM(14);                                   // slot (kt, 0)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) wf[sb] = *(const bf8*)(sw + foff[sb]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            fill_w(kn, par, j);                           // slot (kt, j) is in registers: refill it with k-step kt + 2
            if (j < 2) {
                RF_WAIT_VM(14);                           // slot (kt, j + 1)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) wn[sb] = *(const bf8*)(sw + (uint32_t)((j + 1) * kR2Slot) + foff[sb]);
            }
#if defined(CQS_R2_ABLATE_NOMFMA)   // timing experiment (wrong results)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[j][t][0] += (float)wf[0][t] * (float)af[t][1][0] + (float)wf[1][t & 3] * (float)af[t][0][1];
#else
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[sb], af[t][sb], acc[j][t], 0, 0, 0);
#endif
            if (j < 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wf[0] = wn[0]; wf[1] = wn[1];
            }
        }
    };
    // prologue = the issue pattern of steps -2 and -1
    fill_a(0u, 0u);
#pragma unroll
    for (int j = 0; j < 3; ++j) fill_w(0u, 0u, j);
    {
        const uint32_t k1 = klast >= 1u ? 1u : 0u;
        fill_a(k1, 1u);
#pragma unroll
        for (int j = 0; j < 3; ++j) fill_w(k1, 1u, j);
    }
    {
        uint32_t abuf = 0u;
        for (uint32_t kt = 0; kt <= klast; ++kt) {
            kstep(kt, abuf);
            abuf = abuf == 2u ? 0u : abuf + 1u;
        }
    }
    RF_WAIT_VM(0);                                        // the tail's surplus requests are done before LDS is reused
#if defined(CQS_R2_ABLATE_NOEPI)    // timing experiment (wrong results): the main loop alone
    if (acc[0][0][0] != 12345.678f || acc[1][3][1] != 1.f || acc[2][7][2] != 2.f) return;
#endif

    // ---- epilogue ----
    __syncthreads();                                      // every wave is out of the ring
    bf16_t* const sY = (bf16_t*)rf_smem;                  // [128][kR2Ldy]
    // acc[j][t][r] = y[token 16 t + l15][column n0 + 48 wid + 16 j + 4 lg + r]
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bf4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[j][t][r];
            *(bf4*)(sY + (size_t)(16 * t + l15) * kR2Ldy + 48 * wid + 16 * j + 4 * lg) = o;
        }
    // wave w: rows 16 w .. 16 w + 15, two at a time: lanes 0-31 row 2 i, lanes 32-63 row 2 i + 1; lane (l = lane & 31) owns
    // columns n0 + 128 c + 4 l + {0..3}, c = 0, 1, 2.
    const int hw = lane >> 5, l31 = lane & 31;
    const uint32_t cbase = n0 + (uint32_t)l31 * 4u;
    __syncthreads();
    // the x rows are requested here: their latency hides under phase 1 (LDS only) and the first exchange
    f4 xv[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r = (uint32_t)(16 * wid + 2 * i + hw);
        const uint32_t row = m0 + (r < rows_left ? r : rows_left - 1u);
#pragma unroll
        for (int c = 0; c < 3; ++c) xv[i][c] = *(const f4*)(x + (size_t)row * kRfH + cbase + c * 128);
    }
    f4 wpo[3], wne[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        wpo[c] = *(const f4*)(w_post + cbase + c * 128);
        wne[c] = *(const f4*)(w_next + cbase + c * 128);
    }

    // granules of this wave's 16 rows: lanes with l31 < 8 hold row 2 l31 + hw (= the row of iteration i = l31 in its half-wave)
    r2_gu64* const gx = (r2_gu64*)xch;
    const uint32_t grow = m0 + (uint32_t)(16 * wid + 2 * l31 + hw);               // (< Mpad for l31 < 8)
    auto exchange = [&](uint32_t phase, float mine) -> float {                     // returns the partner's value of the same row
        float theirs = 0.f;
        if (l31 < 8) {
            r2_gu64* const mine_p = gx + ((size_t)(phase * 2u + half) * Mpad + grow);
            r2_gu64* const their_p = gx + ((size_t)(phase * 2u + (half ^ 1u)) * Mpad + grow);
            __hip_atomic_store(mine_p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(mine),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0;
#if defined(CQS_R2_ABLATE_NOEXCH)   // timing experiment (wrong results): nobody waits for the partner
            theirs = mine;
            if (tag == 0u)
#endif
            for (;;) {
                const unsigned long long g = __hip_atomic_load(their_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(g >> 32) == tag) { theirs = __uint_as_float((uint32_t)g); break; }
                if (++spins > kR2SpinMax) {               // the partner never came: never hang, say so
                    __hip_atomic_store((r2_gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        return theirs;
    };
    auto half_sum = [&](float v) -> float { return half_wave_sum32(v); };   // over the 32 lanes of the half-wave (same bits in all of them)

    // phase 1: this half's sum of squares of every y row
    float mine1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r = (uint32_t)(16 * wid + 2 * i + hw);
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bf4 yb = *(const bf4*)(sY + (size_t)r * kR2Ldy + c * 128 + l31 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float v = (float)yb[e]; ss += v * v; }
        }
        ss = half_sum(ss);
        mine1 = (l31 == i) ? ss : mine1;
    }
    const float theirs1 = exchange(0u, mine1);
    // phase 2: x += rmsnorm(y) (1 + w_post); this half's sum of squares of the new x rows
    float mine2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r = (uint32_t)(16 * wid + 2 * i + hw);
        const float a = __shfl(mine1, (lane & 32) + i, 64), b = __shfl(theirs1, (lane & 32) + i, 64);
        const float tot = half == 0u ? a + b : b + a;     // left + right
        const float invy = rsqrtf(tot / (float)kRfH + eps);
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bf4 yb = *(const bf4*)(sY + (size_t)r * kR2Ldy + c * 128 + l31 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[i][c][e] += (float)yb[e] * invy * (1.0f + wpo[c][e]);
                sx += xv[i][c][e] * xv[i][c][e];
            }
            if (r < rows_left) *(f4*)(x + (size_t)(m0 + r) * kRfH + cbase + c * 128) = xv[i][c];
        }
        sx = half_sum(sx);
        mine2 = (l31 == i) ? sx : mine2;
    }
    const float theirs2 = exchange(1u, mine2);
    // phase 3: the next pre-norm of the new x
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r = (uint32_t)(16 * wid + 2 * i + hw);
        const float a = __shfl(mine2, (lane & 32) + i, 64), b = __shfl(theirs2, (lane & 32) + i, 64);
        const float tot = half == 0u ? a + b : b + a;
        const float invx = rsqrtf(tot / (float)kRfH + eps);
        if (r >= rows_left) continue;
        const size_t row = (size_t)(m0 + r);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (FINAL) {
                f4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = xv[i][c][e] * invx * (1.0f + wne[c][e]);
                *(f4*)(out + row * kRfH + cbase + c * 128) = o;
            } else {
                bf4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(xv[i][c][e] * invx * (1.0f + wne[c][e]));
                *(bf4*)(xn + row * kRfH + cbase + c * 128) = o;
            }
        }
    }
}

}  // namespace

bool gemm_addnorm_supported(uint32_t M, uint32_t H, uint32_t K) {   // shape only; whether the forward uses it is the engine's setting
    return M > 0 && H == (uint32_t)kRfH && K % 64u == 0 && K >= 64u && (uint64_t)M * K < (1ull << 31) && (uint64_t)H * K < (1ull << 31);
}

hipError_t launch_gemm_addnorm(const bf16_t* A, const bf16_t* W, float* x, const float* w_post, const float* w_next,
                               float eps, bf16_t* xn, float* out, int final, uint32_t M, uint32_t H, uint32_t K,
                               hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (H != (uint32_t)kRfH || K % 64u || K < 64u || (uint64_t)M * K >= (1ull << 31)) return hipErrorInvalidValue;
    const dim3 grid((M + (uint32_t)kRfRows - 1u) / (uint32_t)kRfRows);
    static DynLdsOnce once[2];
    if (final) {
        const hipError_t e = once[1].ensure((const void*)gemm_rowfuse_kernel<1>, (size_t)kRfLds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gemm_rowfuse_kernel<1>, grid, dim3(512), (size_t)kRfLds, st, A, W, x, w_post, w_next, eps, xn, out, M, K);
    } else {
        const hipError_t e = once[0].ensure((const void*)gemm_rowfuse_kernel<0>, (size_t)kRfLds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gemm_rowfuse_kernel<0>, grid, dim3(512), (size_t)kRfLds, st, A, W, x, w_post, w_next, eps, xn, out, M, K);
    }
    return hipGetLastError();
}

hipError_t launch_pack_rowfuse_w(const bf16_t* src, bf16_t* dst, uint32_t N, uint32_t K, hipStream_t st) {
    if (N % 16u || K % 64u || !src || !dst) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_rowfuse_w_kernel, dim3((N * (K / 8u) + 255u) / 256u), dim3(256), 0, st, src, dst, N, K);
    return hipGetLastError();
}

// The pair-split kernel (gemm_rowfuse2_kernel).  Wp = the weights packed by launch_pack_rowfuse_w.  xch: device buffer of 4 * Mpad u64 granules, Mpad = M rounded up to
// 128, private to the calling stream (launches that share it must be stream-ordered); tag: a number that differs from
// every earlier launch on that buffer (never 0); err: a word the kernel sets to 1 if a pair exchange timed out.
size_t gemm_addnorm_pair_scratch_bytes(uint32_t M) { return (size_t)4 * ((M + 127u) / 128u * 128u) * sizeof(unsigned long long); }

hipError_t launch_gemm_addnorm_pair(const bf16_t* A, const bf16_t* Wp, float* x, const float* w_post, const float* w_next,
                                    float eps, bf16_t* xn, float* out, int final, uint32_t M, uint32_t H, uint32_t K,
                                    void* xch, uint32_t tag, unsigned* err, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (H != (uint32_t)kRfH || K % 64u || K < 64u || (uint64_t)M * K >= (1ull << 31) || !xch || !err || tag == 0u) return hipErrorInvalidValue;
    const uint32_t pairs = (M + (uint32_t)kR2Rows - 1u) / (uint32_t)kR2Rows;
    const uint32_t Mpad = pairs * (uint32_t)kR2Rows;
    // workgroup index = 16 (pair / 8) + 8 half + pair % 8: groups of 16 hold 8 whole pairs; a tail group is sized as a whole
    const dim3 grid((pairs + 7u) / 8u * 16u);
    static DynLdsOnce once[2];
    if (final) {
        const hipError_t e = once[1].ensure((const void*)gemm_rowfuse2_kernel<1>, (size_t)kR2Lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gemm_rowfuse2_kernel<1>, grid, dim3(512), (size_t)kR2Lds, st, A, Wp, x, w_post, w_next, eps, xn, out, M, K,
                           (unsigned long long*)xch, Mpad, tag, err);
    } else {
        const hipError_t e = once[0].ensure((const void*)gemm_rowfuse2_kernel<0>, (size_t)kR2Lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gemm_rowfuse2_kernel<0>, grid, dim3(512), (size_t)kR2Lds, st, A, Wp, x, w_post, w_next, eps, xn, out, M, K,
                           (unsigned long long*)xch, Mpad, tag, err);
    }
    return hipGetLastError();
}

}  // namespace cqs
