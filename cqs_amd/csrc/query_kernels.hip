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
//   * 5 launches per layer, the minimum the data flow allows with every projection's N range spread over the chip (a
//     row norm needs the whole row = all N-slices of the producing GEMM, so every GEMM output is a grid-wide seam; a
//     grid barrier inside one launch costs 4-5 us on this chip, 3x a launch boundary - not a persistent kernel):
//         QKV      [ x += norm(down_prev)(1+w) ; xn = norm(x)(1+w) ]  -> qkv = xn Wqkv^T
//         attention[ k norm + rope, q norm + rope + scale ]           -> softmax(q k^T) v         (one workgroup per q head)
//         o_proj                                                        -> y = attn Wo^T
//         GeGLU    [ x += norm(y)(1+w) ; xn = norm(x)(1+w) ]          -> h = gelu(xn Wg^T) * (xn Wu^T)
//         down                                                          -> y = h Wd^T
//     the row-wise add + RMSNorm pairs live in the PROLOGUE of the GEMM that consumes them: every workgroup recomputes
//     them for all <= 64 rows (<= 150 KB of L2 reads, ~0.3 us of VALU) instead of a launch + a round trip each; workgroup 0
//     also writes the new residual stream to the other of two x buffers (everybody reads the old one: no race);
//   * a GEMM workgroup = 16 output columns x all rows, its 4 waves split K: N / 16 workgroups (48-144) stream one
//     16 x K weight slice each (24-37 KB), every weight fragment requested BEFORE the prologue so that the weight
//     latency hides under the activation round trip + norm; v_mfma_f32_16x16x32_bf16 with the weight rows as the A
//     operand (a lane ends with 4 consecutive output columns of one token row), partial tiles summed through LDS;
//   * the head (final add + norm, mean pool, Dense 768 -> 3072 -> 768) is two more launches of the same kernel;
//   * the sequence length T lives in DEVICE memory (meta[0]): grids do not depend on it, so one captured hipGraph
//     serves every query (embedder.hip).
// K order differs from the batch kernels (4-way K split) -> results agree with the batch path to bf16 rounding noise
// (cosine >= 0.9999, tests/test_query_path_gpu.py), not bit for bit.
#include "embed_kernels.h"
#include "launch_util.h"

#include <cmath>

namespace cqs {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kQfRows = 64;          // max tokens
constexpr int kQfPad = 8;            // bf16 elements of row padding in the LDS activation tile

__device__ __forceinline__ float qf_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float qf_gelu_tanh(float x) {      // as gelu_tanh (embed_kernels.hip)
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = 2.0f * k0 * (x + k1 * x * x * x);
    return x * __frcp_rn(1.0f + __expf(-u2));
}

enum { QF_PRO_NONE = 0, QF_PRO_EMBED = 1, QF_PRO_ADDNORM = 2, QF_PRO_POOL = 3 };
enum { QF_EPI_BF16 = 0, QF_EPI_GEGLU = 1, QF_EPI_F32 = 2 };

struct QfGemmParams {
    const int32_t* meta;      // [0] = T (1..64), [1 + i] = token id of row i
    // prologue inputs
    const bf16_t* emb;        // EMBED: token table [vocab, H]
    float scale;              // EMBED: sqrt(H)
    const float* x_in;        // ADDNORM / POOL: residual stream [64, H] f32
    const bf16_t* y;          // ADDNORM / POOL: branch output [64, H] bf16
    const float* w_post;      // ADDNORM / POOL: post-branch norm weight [H]
    const float* w_next;      // next pre-norm weight [H] (POOL: the model's final norm)
    float* x_out;             // EMBED / ADDNORM: new residual stream [64, H] (written by workgroup 0; != x_in)
    float eps;
    // GEMM
    const bf16_t* A;          // PRO_NONE: activations [64, K] bf16
    const bf16_t* W;          // [N, K] bf16
    void* C;                  // [64, ldc] bf16 / f32 (POOL and its successor: one row)
    uint32_t K, ldc;
    int32_t one_row;          // 1: the activations are ONE row (the pooled vector), whatever T says
};

// C[rows, 16 (x2 for GeGLU)] of one workgroup.  NCH = H / 256 (prologue variants: K = H).
template <int NCH, int PRO, int EPI>
__global__ __launch_bounds__(256) void qf_gemm_kernel(const QfGemmParams p) {
    constexpr int H = NCH * 256;
    constexpr int NT = EPI == QF_EPI_GEGLU ? 2 : 1;               // 16-row weight tiles per workgroup
    constexpr int LDA = H + kQfPad;                                // LDS activation row stride (elements)
    extern __shared__ __attribute__((aligned(16))) unsigned char qf_smem[];
    bf16_t* const sA = (bf16_t*)qf_smem;                           // [64][LDA] (prologue variants only)
    float* const red = (float*)(qf_smem + (PRO != QF_PRO_NONE ? (size_t)kQfRows * LDA * sizeof(bf16_t) : 0));   // [4 waves][NT][4 mt][64 lanes] f4
    float* const pool = red + 4 * NT * 4 * 64 * 4;                 // POOL: [4 waves][H] column sums

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const uint32_t T = (uint32_t)p.meta[0];
    const uint32_t rows = p.one_row ? 1u : T;
    const uint32_t mtiles = (rows + 15u) / 16u;
    const uint32_t K = PRO != QF_PRO_NONE ? (uint32_t)H : p.K;
    const uint32_t kw = K / 4u;                                    // this wave's K range: [wid kw, (wid + 1) kw)
    const uint32_t steps = kw / 32u;

    // weight rows of the workgroup's tiles.  GeGLU: W rows are interleaved per 64 (32 gate rows, then the same
    // channels' 32 up rows; embedder.hip set_tensor) -> channels [16 b, 16 b + 16) = gate rows 64 (b / 2) + 16 (b % 2) + r.
    uint32_t wrow[NT];
    if (EPI == QF_EPI_GEGLU) {
        wrow[0] = 64u * (blockIdx.x >> 1) + 16u * (blockIdx.x & 1u);
        wrow[NT - 1] = wrow[0] + 32u;
    } else {
        wrow[0] = blockIdx.x * 16u;
    }
    const bf16_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = p.W + (size_t)(wrow[t] + (uint32_t)l15) * K + (size_t)wid * kw + 8u * (uint32_t)lg;

    f4 acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = (f4)(0.f);

    if constexpr (PRO != QF_PRO_NONE) {
        // ---- weights first: the wave's whole K range of its tile(s) is 2 NCH fragments per tile, requested now ----
        constexpr int S = 2 * NCH;                                  // k-steps of 32 per wave (H / 4 / 32)
        bf8 wf[NT][S];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < S; ++s) wf[t][s] = *(const bf8*)(wp[t] + 32 * s);

        // ---- prologue: rows wid, wid + 4, ... of the activation tile; lane owns 4 consecutive floats of each 256-chunk ----
        f4 psum[NCH];                                               // POOL: this wave's column sums
#pragma unroll
        for (int c = 0; c < NCH; ++c) psum[c] = (f4)(0.f);
        constexpr int RB = 4;                                       // rows in flight per wave
        for (uint32_t r0 = (uint32_t)wid; r0 < T; r0 += 4u * RB) {
            f4 xv[RB][NCH];
            f4 yv[RB][NCH];
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const uint32_t row = r0 + 4u * (uint32_t)b < T ? r0 + 4u * (uint32_t)b : T - 1u;   // clamped: loaded, not used
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const uint32_t col = (uint32_t)c * 256u + (uint32_t)lane * 4u;
                    if (PRO == QF_PRO_EMBED) {
                        const bf4 e = *(const bf4*)(p.emb + (size_t)(uint32_t)p.meta[1u + row] * H + col);
#pragma unroll
                        for (int i = 0; i < 4; ++i) xv[b][c][i] = (float)e[i] * p.scale;
                    } else {
                        xv[b][c] = *(const f4*)(p.x_in + (size_t)row * H + col);
                        const bf4 yb = *(const bf4*)(p.y + (size_t)row * H + col);
#pragma unroll
                        for (int i = 0; i < 4; ++i) yv[b][c][i] = (float)yb[i];
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const uint32_t row = r0 + 4u * (uint32_t)b;
                if (row >= T) break;                                 // wave-uniform
                if (PRO != QF_PRO_EMBED) {                           // x += norm(y) (1 + w_post)   (add_norm_kernel's arithmetic)
                    float ss = 0.f;
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int i = 0; i < 4; ++i) ss += yv[b][c][i] * yv[b][c][i];
                    const float invy = rsqrtf(qf_wave_sum(ss) / (float)H + p.eps);
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const f4 w = *(const f4*)(p.w_post + c * 256 + lane * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) xv[b][c][i] += yv[b][c][i] * invy * (1.0f + w[i]);
                    }
                }
                float sx = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sx += xv[b][c][i] * xv[b][c][i];
                const float invx = rsqrtf(qf_wave_sum(sx) / (float)H + p.eps);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const uint32_t col = (uint32_t)c * 256u + (uint32_t)lane * 4u;
                    const f4 w = *(const f4*)(p.w_next + col);
                    if (PRO == QF_PRO_POOL) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) psum[c][i] += xv[b][c][i] * invx * (1.0f + w[i]);    // f32 hidden state, summed
                    } else {
                        bf4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(xv[b][c][i] * invx * (1.0f + w[i]));
                        *(bf4*)(sA + (size_t)row * LDA + col) = o;
                        if (blockIdx.x == 0) *(f4*)(p.x_out + (size_t)row * H + col) = xv[b][c];
                    }
                }
            }
        }
        if (PRO == QF_PRO_POOL) {
            // masked mean pool (src/embedder/pooling.rs:87-128) of the final-norm rows -> ONE activation row (bf16, as
            // mean_pool_kernel hands it to the Dense head)
#pragma unroll
            for (int c = 0; c < NCH; ++c) *(f4*)(pool + (size_t)wid * H + c * 256 + lane * 4) = psum[c];
            __syncthreads();
            for (uint32_t col = (uint32_t)tid; col < (uint32_t)H; col += 256u) {
                const float s = pool[col] + pool[H + col] + pool[2 * H + col] + pool[3 * H + col];
                sA[col] = (bf16_t)(s / (float)T);
            }
            for (uint32_t i = (uint32_t)tid; i < 15u * (uint32_t)H; i += 256u) sA[(size_t)(1u + i / H) * LDA + i % H] = (bf16_t)0.f;   // rows 1..15 of the m-tile
        } else {
            // rows [T, 16 mtiles) feed the MFMA's B operand too: keep them finite
            for (uint32_t i = (uint32_t)tid; i < (mtiles * 16u - T) * (uint32_t)H; i += 256u) sA[(size_t)(T + i / H) * LDA + i % H] = (bf16_t)0.f;
        }
        __syncthreads();
        // ---- multiply: B operand (activations) from LDS: lane feeds row 16 m + l15, k = wid kw + 32 s + 8 lg .. + 7 ----
        const bf16_t* la = sA + (size_t)l15 * LDA + (size_t)wid * kw + 8u * (uint32_t)lg;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if ((uint32_t)m >= mtiles) break;                        // wave-uniform
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const bf8 a = *(const bf8*)(la + (size_t)(16 * m) * LDA + 32 * s);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][s], a, acc[t][m], 0, 0, 0);
            }
        }
    } else {
        // ---- activations straight from global memory (o_proj, down, Dense 2): U k-steps of both operands in flight ----
        constexpr int U = 6;
        const bf16_t* ap[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint32_t r = 16u * (uint32_t)m + (uint32_t)l15;
            ap[m] = p.A + (size_t)(r < rows ? r : rows - 1u) * K + (size_t)wid * kw + 8u * (uint32_t)lg;   // rows past the end: any real row
        }
        for (uint32_t s0 = 0; s0 < steps; s0 += U) {
            bf8 wf[NT][U], af[4][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t s = s0 + (uint32_t)u < steps ? s0 + (uint32_t)u : steps - 1u;            // (tail: re-read, not accumulated)
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[t][u] = *(const bf8*)(wp[t] + (size_t)s * 32u);
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if ((uint32_t)m < mtiles) af[m][u] = *(const bf8*)(ap[m] + (size_t)s * 32u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + (uint32_t)u >= steps) break;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if ((uint32_t)m >= mtiles) break;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][u], af[m][u], acc[t][m], 0, 0, 0);
                }
            }
        }
    }

    // ---- sum the four K-quarters through LDS; wave w finishes m-tile w.  acc[t][m][r] = C[row 16 m + l15][col 4 lg + r] ----
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if ((uint32_t)m < mtiles) *(f4*)(red + ((size_t)((wid * NT + t) * 4 + m) * 64 + lane) * 4) = acc[t][m];
    __syncthreads();
    if ((uint32_t)wid >= mtiles) return;
    f4 v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        v[t] = *(const f4*)(red + ((size_t)((0 * NT + t) * 4 + wid) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v[t] += *(const f4*)(red + ((size_t)((w * NT + t) * 4 + wid) * 64 + lane) * 4);
    }
    const uint32_t row = 16u * (uint32_t)wid + (uint32_t)l15;
    if (row >= rows) return;
    if (EPI == QF_EPI_GEGLU) {
        bf4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(qf_gelu_tanh(v[0][r]) * v[NT - 1][r]);
        *(bf4*)((bf16_t*)p.C + (size_t)row * p.ldc + blockIdx.x * 16u + 4u * (uint32_t)lg) = o;
    } else if (EPI == QF_EPI_F32) {
        *(f4*)((float*)p.C + (size_t)row * p.ldc + blockIdx.x * 16u + 4u * (uint32_t)lg) = v[0];
    } else {
        bf4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[0][r];
        *(bf4*)((bf16_t*)p.C + (size_t)row * p.ldc + blockIdx.x * 16u + 4u * (uint32_t)lg) = o;
    }
}

// ---- attention over <= 64 keys: one workgroup per q head, wave w = queries [16 w, 16 w + 16) ---------------------------
// K rows (k-head RMSNorm + RoPE applied on the way, the arithmetic of qk_norm_rope_block) and V^T are staged once in LDS;
// S^T = K Q^T with keys on MFMA rows (softmax lane-local + two cross-lane steps), O^T = V^T P^T with the S^T
// accumulators as the B operand (embed_kernels.hip's key permutation: lane group g holds keys {4g..4g+3} of each 16-key
// tile, so the 8 slots of a 32-key step are keys {4g.., 16 + 4g..} and V^T is read in that order).
constexpr int kQfHD = 256;
constexpr int kQfKRow = kQfHD + 8;          // sK row stride (elements)
constexpr int kQfVRow = kQfRows + 8;        // sVt row stride (elements)

struct QfAttnParams {
    const int32_t* meta;
    const bf16_t* qkv;        // [64, (heads + 2 kv) 256] bf16, as the QKV projection wrote it
    bf16_t* out;              // [64, heads 256] bf16
    const float* wq;          // q-head norm weight [256]
    const float* wk;          // k-head norm weight [256]
    const float* cos_sin;     // [max_seq][128][2] of the layer type
    float eps, q_scale;
    uint32_t heads, kv_heads;
    uint32_t window;          // 0 = full attention; else |q - k| < window
};

__global__ __launch_bounds__(256) void qf_attention_kernel(const QfAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qf_smem[];
    bf16_t* const sK = (bf16_t*)qf_smem;                                  // [64][kQfKRow]
    bf16_t* const sVt = sK + (size_t)kQfRows * kQfKRow;                   // [256][kQfVRow]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const uint32_t T = (uint32_t)p.meta[0];
    const uint32_t mtiles = (T + 15u) / 16u;
    const uint32_t h = blockIdx.x, g = h / (p.heads / p.kv_heads);
    const uint32_t ld = (p.heads + 2u * p.kv_heads) * (uint32_t)kQfHD;
    const bf16_t* kbase = p.qkv + (size_t)(p.heads + g) * kQfHD;
    const bf16_t* vbase = p.qkv + (size_t)(p.heads + p.kv_heads + g) * kQfHD;

    // ---- stage K (norm + rope) and V^T: wave w takes keys w, w + 4, ...; lane owns dims [4 lane, 4 lane + 4) ----
    const f4 wkv = *(const f4*)(p.wk + lane * 4);
    for (uint32_t key = (uint32_t)wid; key < mtiles * 16u; key += 4u) {
        if (key < T) {
            const bf4 kin = *(const bf4*)(kbase + (size_t)key * ld + lane * 4);
            const bf4 vin = *(const bf4*)(vbase + (size_t)key * ld + lane * 4);
            const float* cs = p.cos_sin + ((size_t)key * 128u + (uint32_t)(lane & 31) * 4u) * 2u;
            const f4 cs0 = *(const f4*)cs, cs1 = *(const f4*)(cs + 4);
            const float c4[4] = {cs0[0], cs0[2], cs1[0], cs1[2]};
            const float s4[4] = {cs0[1], cs0[3], cs1[1], cs1[3]};
            float kv[4];
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { kv[i] = (float)kin[i]; ss += kv[i] * kv[i]; }
            const float inv = rsqrtf(qf_wave_sum(ss) / (float)kQfHD + p.eps);
            bf4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float n = kv[i] * inv * (1.0f + wkv[i]);
                const float other = __shfl_xor(n, 32, 64);            // rotate_half partner: dim +/- 128 = lane ^ 32
                o[i] = (bf16_t)((lane < 32) ? (n * c4[i] - other * s4[i]) : (n * c4[i] + other * s4[i]));
            }
            *(bf4*)(sK + (size_t)key * kQfKRow + lane * 4) = o;
#pragma unroll
            for (int i = 0; i < 4; ++i) sVt[(size_t)(lane * 4 + i) * kQfVRow + key] = vin[i];
        } else {                                                      // padding keys: finite (their P is 0)
            *(bf4*)(sK + (size_t)key * kQfKRow + lane * 4) = (bf4)(0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) sVt[(size_t)(lane * 4 + i) * kQfVRow + key] = (bf16_t)0.f;
        }
    }
    // a 32-key PV step past the last 16-key tile reads 16 more V^T columns: zero them too
    if (mtiles & 1u)
        for (uint32_t i = (uint32_t)tid; i < 16u * (uint32_t)kQfHD; i += 256u) sVt[(size_t)(i >> 4) * kQfVRow + mtiles * 16u + (i & 15u)] = (bf16_t)0.f;

    // ---- this wave's Q^T fragments (B operand of S^T): q norm + rope + scale on the wave's own fragments ----
    const bool active = (uint32_t)wid < mtiles;
    bf8 qf[8];
    const uint32_t q = 16u * (uint32_t)wid + (uint32_t)l15;
    const uint32_t qc = q < T ? q : T - 1u;
    if (active) {
        const bf16_t* qrow = p.qkv + (size_t)qc * ld + (size_t)h * kQfHD;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *(const bf8*)(qrow + 32 * s + 8 * lg);
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float v = (float)qf[s][j]; ss += v * v; }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float inv = rsqrtf(ss / (float)kQfHD + p.eps);
        const float* cs = p.cos_sin + (size_t)qc * 256u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t d0 = (uint32_t)(32 * s + 8 * lg);        // dims d0 .. d0 + 7 (< 128) and their partners d0 + 128
            f4 c[4], wlo[2], whi[2];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = *(const f4*)(cs + 2u * d0 + 4u * (uint32_t)u);
#pragma unroll
            for (int u = 0; u < 2; ++u) { wlo[u] = *(const f4*)(p.wq + d0 + 4 * u); whi[u] = *(const f4*)(p.wq + 128u + d0 + 4 * u); }
            bf8 lo, hi;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float co = c[j >> 1][2 * (j & 1)], si = c[j >> 1][2 * (j & 1) + 1];
                const float nlo = (float)qf[s][j] * inv * (1.0f + wlo[j >> 2][j & 3]);
                const float nhi = (float)qf[s + 4][j] * inv * (1.0f + whi[j >> 2][j & 3]);
                lo[j] = (bf16_t)((nlo * co - nhi * si) * p.q_scale);
                hi[j] = (bf16_t)((nhi * co + nlo * si) * p.q_scale);
            }
            qf[s] = lo;
            qf[s + 4] = hi;
        }
    }
    __syncthreads();
    if (!active) return;

    // ---- S^T tiles: sc[kt][r] = S[key 16 kt + 4 lg + r][query l15] ----
    f4 sc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        sc[kt] = (f4)(0.f);
        if ((uint32_t)kt >= mtiles) continue;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf8 kf = *(const bf8*)(sK + (size_t)(16 * kt + l15) * kQfKRow + 32 * s + 8 * lg);
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], sc[kt], 0, 0, 0);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t key = (uint32_t)(16 * kt + 4 * lg + r);
            const uint32_t dist = key > q ? key - q : q - key;
            const bool ok = key < T && (p.window == 0u || dist < p.window);
            sc[kt][r] = ok ? sc[kt][r] : -INFINITY;
            mx = fmaxf(mx, sc[kt][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    bf4 pb[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __expf(sc[kt][r] - mx);                   // masked: exp(-inf) = 0 (a query always sees itself: mx is finite)
            pb[kt][r] = (bf16_t)e;
            sum += (float)pb[kt][r];                                   // the rounded weights are what multiplies V
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float rinv = 1.0f / sum;

    // ---- O^T = V^T P^T over 32-key steps; o[dt][r] = O[query l15][dim 16 dt + 4 lg + r] ----
    f4 o[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) o[dt] = (f4)(0.f);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if ((uint32_t)(2 * kb) >= mtiles) break;
        bf8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = pb[2 * kb][r]; pf[4 + r] = pb[2 * kb + 1][r]; }
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            const bf16_t* vr = sVt + (size_t)(16 * dt + l15) * kQfVRow + 32 * kb + 4 * lg;
            const bf4 v0 = *(const bf4*)vr, v1 = *(const bf4*)(vr + 16);
            bf8 vf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { vf[r] = v0[r]; vf[4 + r] = v1[r]; }
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        }
    }
    if (q >= T) return;
    bf16_t* orow = p.out + (size_t)q * (p.heads * (uint32_t)kQfHD) + (size_t)h * kQfHD + 4 * lg;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
        bf4 ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)(o[dt][r] * rinv);
        *(bf4*)(orow + 16 * dt) = ob;
    }
}

template <int NCH, int PRO, int EPI>
hipError_t qf_launch_gemm_t(const QfGemmParams& p, uint32_t n_out_cols, hipStream_t st) {
    constexpr int NT = EPI == QF_EPI_GEGLU ? 2 : 1;
    size_t lds = (size_t)4 * NT * 4 * 64 * 16;                                     // K-split partial tiles
    if (PRO != QF_PRO_NONE) lds += (size_t)kQfRows * (NCH * 256 + kQfPad) * sizeof(bf16_t);
    if (PRO == QF_PRO_POOL) lds += (size_t)4 * NCH * 256 * sizeof(float);
    auto kern = qf_gemm_kernel<NCH, PRO, EPI>;
    static DynLdsOnce once;
    const hipError_t e = once.ensure((const void*)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_out_cols / 16u), dim3(256), lds, st, p);
    return hipGetLastError();
}

template <int NCH>
hipError_t qf_forward_t(const QueryFwd& f, hipStream_t st) {
    const uint32_t H = f.hidden, NQ = (f.heads + 2u * f.kv_heads) * 256u, HQ = f.heads * 256u;
    float* xb[2] = {f.x0, f.x1};
    int cur = 0;                                     // the residual stream lives in xb[cur]
    const size_t att_lds = ((size_t)kQfRows * kQfKRow + (size_t)kQfHD * kQfVRow) * sizeof(bf16_t);
    static DynLdsOnce att_once;
    hipError_t e = att_once.ensure((const void*)qf_attention_kernel, att_lds);
    if (e != hipSuccess) return e;
    for (uint32_t l = 0; l < f.layers; ++l) {
        const QueryFwdLayer& w = f.layer[l];
        QfGemmParams p{};
        p.meta = f.meta; p.eps = f.eps;
        // QKV (+ embedding gather / the previous layer's post-ffw add + this layer's input norm)
        p.W = w.wqkv; p.C = f.qkv; p.ldc = NQ; p.w_next = w.n_in; p.x_out = xb[cur ^ (l == 0 ? 0 : 1)];
        if (l == 0) {
            p.emb = f.emb; p.scale = f.embed_scale;
            e = qf_launch_gemm_t<NCH, QF_PRO_EMBED, QF_EPI_BF16>(p, NQ, st);
        } else {
            p.x_in = xb[cur]; p.y = f.y; p.w_post = f.layer[l - 1].n_post_ffw;
            e = qf_launch_gemm_t<NCH, QF_PRO_ADDNORM, QF_EPI_BF16>(p, NQ, st);
            cur ^= 1;
        }
        if (e != hipSuccess) return e;
        // attention
        QfAttnParams a{};
        a.meta = f.meta; a.qkv = f.qkv; a.out = f.attn; a.wq = w.n_q; a.wk = w.n_k;
        const bool full = ((l + 1u) % f.sliding_pattern) == 0u;
        a.cos_sin = full ? f.rope_global : f.rope_local;
        a.eps = f.eps; a.q_scale = f.q_scale; a.heads = f.heads; a.kv_heads = f.kv_heads; a.window = full ? 0u : f.window;
        hipLaunchKernelGGL(qf_attention_kernel, dim3(f.heads), dim3(256), att_lds, st, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // o_proj
        QfGemmParams po{};
        po.meta = f.meta; po.A = f.attn; po.K = HQ; po.W = w.wo; po.C = f.y; po.ldc = H;
        if ((e = qf_launch_gemm_t<NCH, QF_PRO_NONE, QF_EPI_BF16>(po, H, st)) != hipSuccess) return e;
        // GeGLU (+ post-attention add, pre-ffw norm)
        QfGemmParams pg{};
        pg.meta = f.meta; pg.eps = f.eps; pg.x_in = xb[cur]; pg.y = f.y; pg.w_post = w.n_post_attn; pg.w_next = w.n_pre_ffw;
        pg.x_out = xb[cur ^ 1]; pg.W = w.wgu; pg.C = f.h; pg.ldc = f.inter;
        if ((e = qf_launch_gemm_t<NCH, QF_PRO_ADDNORM, QF_EPI_GEGLU>(pg, f.inter, st)) != hipSuccess) return e;
        cur ^= 1;
        // down
        QfGemmParams pd{};
        pd.meta = f.meta; pd.A = f.h; pd.K = f.inter; pd.W = w.wd; pd.C = f.y; pd.ldc = H;
        if ((e = qf_launch_gemm_t<NCH, QF_PRO_NONE, QF_EPI_BF16>(pd, H, st)) != hipSuccess) return e;
    }
    // head: last post-ffw add + final norm + mean pool -> Dense 1 ; Dense 2
    QfGemmParams p1{};
    p1.meta = f.meta; p1.eps = f.eps; p1.x_in = xb[cur]; p1.y = f.y; p1.w_post = f.layer[f.layers - 1].n_post_ffw; p1.w_next = f.n_final;
    p1.W = f.dense1; p1.C = f.d1; p1.ldc = f.dense_hidden; p1.one_row = 1;
    if ((e = qf_launch_gemm_t<NCH, QF_PRO_POOL, QF_EPI_BF16>(p1, f.dense_hidden, st)) != hipSuccess) return e;
    QfGemmParams p2{};
    p2.meta = f.meta; p2.A = f.d1; p2.K = f.dense_hidden; p2.W = f.dense2; p2.C = f.out; p2.ldc = H; p2.one_row = 1;
    return qf_launch_gemm_t<NCH, QF_PRO_NONE, QF_EPI_F32>(p2, H, st);
}

}  // namespace

bool query_forward_supported(const EmbedGeom& g) {
    const uint32_t HQ = g.heads * 256u;
    return (g.hidden == 256u || g.hidden == 768u) && g.head_dim == 256u && g.kv_heads && g.heads % g.kv_heads == 0 &&
           g.inter % 128u == 0 && HQ % 128u == 0 && g.dense_hidden % 128u == 0 && g.hidden % 128u == 0;
}

hipError_t launch_query_forward(const QueryFwd& f, hipStream_t st) {
    if (!f.layer || f.layers == 0) return hipErrorInvalidValue;
    if (f.hidden == 768u) return qf_forward_t<3>(f, st);
    if (f.hidden == 256u) return qf_forward_t<1>(f, st);
    return hipErrorInvalidValue;
}

}  // namespace cqs
