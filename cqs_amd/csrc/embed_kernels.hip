// embed_kernels.hip — gfx950 kernels of the EmbeddingGemma-300m forward (Gemma3 text encoder,
// bidirectional, + sentence-transformers pooling/dense head).  Replaces the ONNX Runtime
// `session.run` of the reference (src/embedder/core.rs:1097; graph described in SURVEY.md
// §8a row A20).  Semantics follow oracle/gemma3_ref.py (which is pinned to transformers'
// Gemma3TextModel); bf16 operands on the matrix cores, f32 accumulation, f32 residual stream.
//
// Tokens are PACKED: padding never reaches a kernel (the reference pads every sequence to the
// longest of the batch, src/embedder/core.rs:1020-1035, and ORT computes on the pad).
#include "embed_kernels.h"

#include <cstdlib>
#include <type_traits>

namespace cqs {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

constexpr int kHD = 256;  // head_dim the attention / rope kernels are specialised for

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- row kernels: one wave per token row, lane owns 4 consecutive floats of each 256-chunk ----
template <int NCH>
__global__ __launch_bounds__(256) void embed_norm_kernel(const int32_t* __restrict__ tok,
                                                         const bf16_t* __restrict__ emb, float scale,
                                                         const float* __restrict__ w_in, float eps,
                                                         float* __restrict__ x, bf16_t* __restrict__ xn, uint32_t M) {
    constexpr uint32_t H = NCH * 256;
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t src = (size_t)tok[row] * H;
    float v[NCH][4];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t col = (uint32_t)c * 256u + (uint32_t)lane * 4u;
        const bf4 e = *(const bf4*)(emb + src + col);
        f4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[c][i] = (float)e[i] * scale;
            o[i] = v[c][i];
            ss += v[c][i] * v[c][i];
        }
        *(f4*)(x + (size_t)row * H + col) = o;
    }
    const float inv = rsqrtf(wave_sum(ss) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t col = (uint32_t)c * 256u + (uint32_t)lane * 4u;
        const f4 w = *(const f4*)(w_in + col);
        bf4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(v[c][i] * inv * (1.0f + w[i]));
        *(bf4*)(xn + (size_t)row * H + col) = o;
    }
}

// Row sums are taken HALF BY HALF: lane L owns, in half h = L >> 5, the columns (H / 2) h + 128 c + 4 (L & 31) + {0..3},
// c = 0..NCH-1; a half's partial is summed over its 32 lanes (butterfly 16, 8, 4, 2, 1) and the row's sum of squares is
// left + right.  That is the order in which the pair-split fused kernel (gemm_rowfuse.hip: two workgroups own the two
// column halves of a row block and exchange their partials) can also sum - so the two stay bit-identical.
__device__ __forceinline__ float half_row_sum(float v) { return half_wave_sum32(v); }   // over the 32 lanes of this lane's half-wave
__device__ __forceinline__ float row_sum_of_halves(float v, int lane) {
    v = half_row_sum(v);
    const lane_u2 a = swap32_self(__float_as_uint(v));   // a[0] = left, a[1] = right in every lane
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);     // left + right on both sides
}
template <int NCH, int FINAL>
__global__ __launch_bounds__(256) void add_norm_kernel(float* __restrict__ x, const bf16_t* __restrict__ y,
                                                       const float* __restrict__ w_post,
                                                       const float* __restrict__ w_next, float eps,
                                                       bf16_t* __restrict__ xn, float* __restrict__ out, uint32_t M) {
    constexpr uint32_t H = NCH * 256;
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= M) return;
    const uint32_t col0 = (uint32_t)(lane >> 5) * (H / 2u) + (uint32_t)(lane & 31) * 4u;
    f4 yv[NCH], xv[NCH];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t col = col0 + (uint32_t)c * 128u;
        const bf4 yb = *(const bf4*)(y + (size_t)row * H + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) yv[c][i] = (float)yb[i];
        xv[c] = *(const f4*)(x + (size_t)row * H + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) ss += yv[c][i] * yv[c][i];
    }
    const float invy = rsqrtf(row_sum_of_halves(ss, lane) / (float)H + eps);
    float sx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t col = col0 + (uint32_t)c * 128u;
        const f4 w = *(const f4*)(w_post + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xv[c][i] += yv[c][i] * invy * (1.0f + w[i]);
            sx += xv[c][i] * xv[c][i];
        }
        *(f4*)(x + (size_t)row * H + col) = xv[c];
    }
    const float invx = rsqrtf(row_sum_of_halves(sx, lane) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t col = col0 + (uint32_t)c * 128u;
        const f4 w = *(const f4*)(w_next + col);
        if (FINAL) {
            f4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = xv[c][i] * invx * (1.0f + w[i]);
            *(f4*)(out + (size_t)row * H + col) = o;
        } else {
            bf4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(xv[c][i] * invx * (1.0f + w[i]));
            *(bf4*)(xn + (size_t)row * H + col) = o;
        }
    }
}

// ---- q/k RMSNorm + RoPE, in place; one wave per token, all its q and k heads; lane owns dims [4l, 4l+4) ----
// (the token's cos/sin row - twice the bytes of one head - is fetched once for all heads, and the heads'
// loads are in flight together)
constexpr int kMaxQkHeads = 8;
__device__ __forceinline__ void qk_norm_rope_block(uint32_t bid, bf16_t* __restrict__ qkv, const int32_t* __restrict__ pos,
                                                   const float* __restrict__ wq, const float* __restrict__ wk,
                                                   const float* __restrict__ cos_sin, float eps, float q_scale,
                                                   uint32_t M, uint32_t heads, uint32_t kv_heads, uint32_t h_first) {
    // heads [h_first, heads + kv_heads) of the fused q | k | v row: h_first = 0 -> q and k heads, h_first = heads -> the
    // k heads only (the attention kernel then normalises / rotates its own Q fragments)
    const int lane = threadIdx.x & 63;
    const uint32_t nh = heads + kv_heads - h_first;
    const uint32_t m = bid * 4u + (threadIdx.x >> 6);
    if (m >= M) return;
    const uint32_t ld = (heads + 2u * kv_heads) * kHD;
    bf16_t* row = qkv + (size_t)m * ld + (size_t)h_first * kHD + lane * 4;
    bf4 in[kMaxQkHeads];
#pragma unroll
    for (int h = 0; h < kMaxQkHeads; ++h)
        if ((uint32_t)h < nh) in[h] = *(const bf4*)(row + (size_t)h * kHD);
    // rotate_half pairs dim d with d +/- 128: the partner lives in lane ^ 32, same element
    const float* cs = cos_sin + ((size_t)pos[m] * 128u + (uint32_t)(lane & 31) * 4u) * 2u;
    const f4 cs0 = *(const f4*)cs, cs1 = *(const f4*)(cs + 4);  // (cos,sin) x 4 dims
    const float c4[4] = {cs0[0], cs0[2], cs1[0], cs1[2]};
    const float s4[4] = {cs0[1], cs0[3], cs1[1], cs1[3]};
    const f4 wqv = *(const f4*)(wq + lane * 4), wkv = *(const f4*)(wk + lane * 4);
#pragma unroll
    for (int h = 0; h < kMaxQkHeads; ++h) {
        if ((uint32_t)h >= nh) break;
        const bool is_q = (uint32_t)h + h_first < heads;
        float v[4];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = (float)in[h][i];
            ss += v[i] * v[i];
        }
        const float inv = rsqrtf(wave_sum(ss) / (float)kHD + eps);
        bf4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float n = v[i] * inv * (1.0f + (is_q ? wqv[i] : wkv[i]));
            const float other = __shfl_xor(n, 32, 64);
            // d < 128: n*cos - x[d+128]*sin ; d >= 128: n*cos + x[d-128]*sin
            float r = (lane < 32) ? (n * c4[i] - other * s4[i]) : (n * c4[i] + other * s4[i]);
            if (is_q) r *= q_scale;
            o[i] = (bf16_t)r;
        }
        *(bf4*)(row + (size_t)h * kHD) = o;
    }
}
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* __restrict__ qkv, const int32_t* __restrict__ pos,
                                                           const float* __restrict__ wq, const float* __restrict__ wk,
                                                           const float* __restrict__ cos_sin, float eps, float q_scale,
                                                           uint32_t M, uint32_t heads, uint32_t kv_heads, uint32_t h_first) {
    qk_norm_rope_block(blockIdx.x, qkv, pos, wq, wk, cos_sin, eps, q_scale, M, heads, kv_heads, h_first);
}

// ---- V transpose: vt[g][d][vt_start[seq] + pos] = v[token][g][d] -------------------------------------
// One workgroup = 64 positions (half a 128-position super-block of the attention's blk list) x 64 head
// dims of one kv head, through LDS.  In: 16-B loads along d.  Out: 16-B stores of 8 consecutive
// positions of one head dim (V^T columns of a sequence start at a multiple of 32, so they are aligned);
// 8 lanes cover one dim's 64 positions = one full 128-B line.  Positions past the sequence end are
// written as zeros (the attention multiplies them by P = 0; they must stay finite).
__device__ __forceinline__ void v_transpose_block(uint32_t bx, uint32_t by, const bf16_t* __restrict__ qkv,
                                                  bf16_t* __restrict__ vt, const int32_t* __restrict__ blk,
                                                  const int32_t* __restrict__ seq_start,
                                                  const int32_t* __restrict__ seq_len,
                                                  const int32_t* __restrict__ vt_start, uint32_t heads,
                                                  uint32_t kv_heads, uint32_t vt_ld) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][64 + 8];
    const uint32_t g = by >> 2, d0 = (by & 3u) * 64u;
    const uint32_t sblk = bx >> 1, half = bx & 1u;
    const uint32_t seq = (uint32_t)blk[2 * sblk], sb = (uint32_t)blk[2 * sblk + 1];
    const uint32_t len = (uint32_t)seq_len[seq], m_seq = (uint32_t)seq_start[seq], c_seq = (uint32_t)vt_start[seq];
    const uint32_t cols = (len + 31u) & ~31u;  // the sequence's padded V^T columns
    const uint32_t p0 = sb * 128u + half * 64u;
    if (p0 >= cols) return;
    const uint32_t ld = (heads + 2u * kv_heads) * kHD;
    const uint32_t voff = (heads + kv_heads + g) * kHD + d0;
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t i = (uint32_t)(u * 256 + tid), t = i >> 3, c = (i & 7u) * 8u;
        bf8 v = (bf8)(0.f);
        if (p0 + t < len) v = *(const bf8*)(qkv + (size_t)(m_seq + p0 + t) * ld + voff + c);
        *(bf8*)&tile[t][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t i = (uint32_t)(u * 256 + tid), d = i >> 3, t0 = (i & 7u) * 8u;
        if (p0 + t0 >= cols) continue;
        bf8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = tile[t0 + e][d];
        *(bf8*)(vt + ((size_t)g * kHD + d0 + d) * vt_ld + c_seq + p0 + t0) = o;
    }
}
__global__ __launch_bounds__(256) void v_transpose_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ vt,
                                                          const int32_t* __restrict__ blk,
                                                          const int32_t* __restrict__ seq_start,
                                                          const int32_t* __restrict__ seq_len,
                                                          const int32_t* __restrict__ vt_start, uint32_t heads,
                                                          uint32_t kv_heads, uint32_t vt_ld) {
    v_transpose_block(blockIdx.x, blockIdx.y, qkv, vt, blk, seq_start, seq_len, vt_start, heads, kv_heads, vt_ld);
}

// ---- k-head norm + RoPE and the V transpose in ONE launch (two 6-7 us latency-bound kernels with nothing in common
// but their input row: workgroups [0, n_rope) take 4 tokens each, the rest one V^T tile each) -------------------------
__global__ __launch_bounds__(256) void kv_prep_kernel(bf16_t* __restrict__ qkv, bf16_t* __restrict__ vt,
                                                      const int32_t* __restrict__ pos, const float* __restrict__ wq,
                                                      const float* __restrict__ wk, const float* __restrict__ cos_sin,
                                                      float eps, float q_scale, uint32_t M, uint32_t heads,
                                                      uint32_t kv_heads, uint32_t n_rope, const int32_t* __restrict__ blk,
                                                      uint32_t nblk, const int32_t* __restrict__ seq_start,
                                                      const int32_t* __restrict__ seq_len,
                                                      const int32_t* __restrict__ vt_start, uint32_t vt_ld) {
    if (blockIdx.x < n_rope) {
        qk_norm_rope_block(blockIdx.x, qkv, pos, wq, wk, cos_sin, eps, q_scale, M, heads, kv_heads, heads);
    } else {
        const uint32_t b2 = blockIdx.x - n_rope;
        v_transpose_block(b2 % (nblk * 2u), b2 / (nblk * 2u), qkv, vt, blk, seq_start, seq_len, vt_start, heads, kv_heads, vt_ld);
    }
}

// ---- attention --------------------------------------------------------------------------
// S^T = K Q^T (keys on rows, queries on lanes: softmax is lane-local), O^T += V^T P^T with the
// S^T accumulator used directly as the B operand (no LDS round trip for P).
// LDS tiles are split by the MFMA lane group g = lane >> 4 so that a fragment read's bank only depends
// on its row:  sK[g][key][8 k-steps x 8 dims + 8 pad]  (row stride 144 B: 16 rows -> 16 distinct 16-B slots)
//              sV[g][dim][8 keys = the group's B-operand slots]  (row stride 16 B: 16 rows = one 256-B bank row)
constexpr int kKRow = 64 + 8;               // elements per sK row
constexpr int kKSub = 32 * kKRow;           // elements per lane-group sub-tile (4608 B = 18 x 256 B)
constexpr int kVSub = kHD * 8;              // elements per lane-group sub-tile of sV

// One workgroup = 8 waves (2 per SIMD: one wave's MFMAs overlap the other's softmax VALU) = 128
// consecutive queries of ONE q-head of one sequence, 16 queries per wave (16x16x32 MFMA tiles keep the
// wave at ~150 VGPRs: O^T 64 + Q 32 + S 8, no accumulator spills into AGPRs).  The K / V^T tiles of
// the head's kv group are staged once per 32-key block for all eight waves.
//   S^T (32 keys x 16 q)  = K (A: 16 keys x 32 dims per tile) x Q^T (B, registers)     16 MFMAs
//   O^T (256 d x 16 q)   += V^T (A: 16 dims x 32 keys) x P^T (B = the S^T registers)    16 MFMAs
// C layout of 16x16: col = lane&15 (query), row = 4*(lane>>4) + reg.  So lane group g = lane>>4 owns
// keys {4g..4g+3} of each 16-key tile; used as the B operand its 8 slots are keys
// {4g..4g+3, 16+4g..16+4g+3} of the block, and the V^T fragment is read in the same order.
constexpr float kRescaleThr = 8.0f;  // defer the O rescale while the running max grows by < e^8 (P stays < 2981)
typedef float f4v __attribute__((ext_vector_type(4)));

// Q^T fragments (B operand of S^T = K Q^T): lane feeds Q[q = lane & 15][dims 32s + 8 lg + 0..7], s = 0..7.
// With q_norm_w: q-head RMSNorm * (1 + w), RoPE and the 1/sqrt(query_pre_attn_scalar) scale on the wave's own
// fragments (what qk_norm_rope_kernel does for the k heads): the query's 256 dims live in this lane (64 of them) and
// in lanes ^16, ^32, ^48; rotate_half pairs dim d with d +/- 128 = fragments s and s + 4 of the SAME lane.
__device__ __forceinline__ void load_q_fragments(bf8 (&qf)[8], const bf16_t* __restrict__ qrow /*token row + head*/, int lg,
                                                 const float* __restrict__ q_norm_w, const float* __restrict__ cs /*[128][2] of the position*/,
                                                 float eps, float q_scale) {
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *(const bf8*)(qrow + 32 * s + 8 * lg);
    if (!q_norm_w) return;
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = (float)qf[s][j]; ss += v * v; }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float inv = rsqrtf(ss / (float)kHD + eps);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint32_t d0 = (uint32_t)(32 * s + 8 * lg);    // dims d0 .. d0+7 (< 128) and their partners d0 + 128
        f4 c[4], wlo[2], whi[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = *(const f4*)(cs + 2u * d0 + 4u * (uint32_t)u);   // (cos,sin) x 2 dims each
#pragma unroll
        for (int u = 0; u < 2; ++u) { wlo[u] = *(const f4*)(q_norm_w + d0 + 4 * u); whi[u] = *(const f4*)(q_norm_w + 128u + d0 + 4 * u); }
        bf8 lo, hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float co = c[j >> 1][2 * (j & 1)], si = c[j >> 1][2 * (j & 1) + 1];
            const float nlo = (float)qf[s][j] * inv * (1.0f + wlo[j >> 2][j & 3]);
            const float nhi = (float)qf[s + 4][j] * inv * (1.0f + whi[j >> 2][j & 3]);
            lo[j] = (bf16_t)((nlo * co - nhi * si) * q_scale);      // d < 128: n cos - x[d+128] sin
            hi[j] = (bf16_t)((nhi * co + nlo * si) * q_scale);      // d >= 128: n cos + x[d-128] sin
        }
        qf[s] = lo;
        qf[s + 4] = hi;
    }
}

template <int WAVES, int G>
__global__ __launch_bounds__(64 * WAVES) void attention_kernel(const bf16_t* __restrict__ qkv,
                                                        const bf16_t* __restrict__ vt,
                                                        bf16_t* __restrict__ out,
                                                        const int32_t* __restrict__ blk,
                                                        const int32_t* __restrict__ seq_start,
                                                        const int32_t* __restrict__ seq_len,
                                                        const int32_t* __restrict__ vt_start, uint32_t vt_ld,
                                                        uint32_t heads, uint32_t kv_heads, uint32_t window,
                                                        const float* __restrict__ q_norm_w,
                                                        const float* __restrict__ cos_sin, float eps, float q_scale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * kKSub + 4 * kVSub];
    bf16_t* sK = smem;
    bf16_t* sV = smem + 4 * kKSub;
    // The workgroup owns TQ = WAVES / G tiles of 16 consecutive queries for ALL G q-heads of one kv head:
    // wave -> (query tile wid / G, head wid % G), so one staged K / V^T tile serves G heads.
    constexpr int T = 64 * WAVES;                   // threads
    constexpr int TQ = WAVES / G;                   // query tiles
    constexpr uint32_t kParts = 128 / (16 * TQ);    // workgroups per 128-query super-block of the blk list
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const uint32_t sblk = blockIdx.x / kParts, part = blockIdx.x % kParts;
    const uint32_t b = (uint32_t)blk[2 * sblk], sb = (uint32_t)blk[2 * sblk + 1];
    // blockIdx.y counts groups of G consecutive q-heads (G = heads / kv_heads: one kv head; G = 1: one q-head)
    const uint32_t head = blockIdx.y * (uint32_t)G + (uint32_t)(wid % G);
    const uint32_t g = head / (heads / kv_heads);
    const uint32_t s0 = (uint32_t)seq_start[b], L = (uint32_t)seq_len[b], v0 = (uint32_t)vt_start[b];
    const uint32_t ld = (heads + 2u * kv_heads) * kHD;
    const uint32_t koff = (heads + g) * kHD;
    const uint32_t qbase = sb * 128u + part * (16u * TQ);      // the workgroup's first query
    if (qbase >= L) return;                                    // (uniform: before any barrier)
    const uint32_t q0 = qbase + (uint32_t)(wid / G) * 16u;     // this wave's first query
    const bool wave_live = q0 < L;                          // waves past the sequence only help staging
    const uint32_t qi = q0 + (uint32_t)l15;                 // this lane's query

    const uint32_t qclamp = qi < L ? qi : L - 1u;           // position in the sequence (rows past the end: any real row)
    bf8 qf[8];
    load_q_fragments(qf, qkv + (size_t)(s0 + qclamp) * ld + head * kHD, lg, q_norm_w, cos_sin + (size_t)qclamp * 256u, eps, q_scale);

    f4v o[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) o[d] = (f4v)(0.f);
    float m_run = -INFINITY, l_run = 0.f;

    // key blocks that can hold an attendable key: workgroup range (staging) and this wave's own range
    const uint32_t nkb = (L + 31u) / 32u;
    uint32_t kb_lo = 0, kb_hi = nkb, wkb_lo = 0, wkb_hi = nkb;
    if (window) {
        const uint32_t glo = qbase, ghi = glo + 16u * TQ - 1u;     // workgroup's queries
        kb_lo = (glo + 1u > window) ? (glo + 1u - window) / 32u : 0u;
        kb_hi = (ghi + window - 1u) / 32u + 1u;
        if (kb_hi > nkb) kb_hi = nkb;
        const uint32_t qhi = q0 + 15u;                              // this wave's queries
        wkb_lo = (q0 + 1u > window) ? (q0 + 1u - window) / 32u : 0u;
        wkb_hi = (qhi + window - 1u) / 32u + 1u;
    }

    // K / V^T tiles go through registers one key block ahead (kU + kU x 16 B per thread)
    constexpr int kU = (1024 + T - 1) / T;
    u4 rk[kU], rv[kU];
    auto stage_load = [&](uint32_t kb) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = u * T + tid;
            if (1024 % T != 0 && i >= 1024) break;
            const uint32_t kr = (uint32_t)i / (kHD / 8), c = ((uint32_t)i % (kHD / 8)) * 8u;
            uint32_t key = kb * 32u + kr;
            key = key < L ? key : L - 1u;   // rows past the sequence: any finite row, masked later
            rk[u] = *(const u4*)(qkv + (size_t)(s0 + key) * ld + koff + c);
            const uint32_t d = (uint32_t)i / 4u, cv = ((uint32_t)i % 4u) * 8u;
            rv[u] = *(const u4*)(vt + ((size_t)g * kHD + d) * vt_ld + v0 + kb * 32u + cv);
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = u * T + tid;
            if (1024 % T != 0 && i >= 1024) break;
            // K: 16-B chunk cc of key row kr = dims 8cc..8cc+7 = k-step cc/4, lane group cc%4
            const uint32_t kr = (uint32_t)i / (kHD / 8), cc = (uint32_t)i % (kHD / 8);
            *(u4*)(sK + (cc & 3u) * kKSub + kr * kKRow + (cc >> 2) * 8u) = rk[u];
            // V^T: chunk a of dim row d = keys 8a..8a+7: keys 8a+0..3 are slots 4(a/2)..+3 of lane group
            // 2(a%2), keys 8a+4..7 the same slots of lane group 2(a%2)+1
            const uint32_t d = (uint32_t)i / 4u, a = (uint32_t)i % 4u;
            bf16_t* vd = sV + (2u * (a & 1u)) * kVSub + d * 8u + 4u * (a >> 1);
            *(u2*)vd = (u2){rv[u][0], rv[u][1]};
            *(u2*)(vd + kVSub) = (u2){rv[u][2], rv[u][3]};
        }
    };
    if (kb_lo < kb_hi) stage_load(kb_lo);
    for (uint32_t kb = kb_lo; kb < kb_hi; ++kb) {
        __syncthreads();  // previous tile fully consumed
        stage_write();
        __syncthreads();
#ifndef CQS_ATT_ABLATE_NOLOAD
        if (kb + 1u < kb_hi) stage_load(kb + 1u);
#endif
        if (!wave_live || kb < wkb_lo || kb >= wkb_hi) continue;   // wave-uniform
#ifdef CQS_ATT_ABLATE_NOCOMPUTE
        if (kb != kb_lo) continue;
#endif

        // S^T tiles: keys [0,16) and [16,32) of the block x this wave's 16 queries.  Fragment reads are
        // issued 8 at a time AHEAD of their MFMAs (left alone, hipcc emits read -> wait -> MFMA pairs and
        // every MFMA eats a full LDS round trip).
        f4v sc[2];
        sc[0] = (f4v)(0.f);
        sc[1] = (f4v)(0.f);
        // kRA fragment reads in flight ahead of their MFMAs: 8 with two waves per SIMD; 4 for the 12-wave layout
        // (three waves per SIMD hide the rest, and 8 spilled 13 registers to scratch at its 168-VGPR budget)
        constexpr int kRA = WAVES >= 12 ? 4 : 8;
#pragma unroll
        for (int grp = 0; grp < 16 / kRA; ++grp) {
            bf8 kf[kRA];
#pragma unroll
            for (int u = 0; u < kRA; ++u) {
                const int i = grp * kRA + u, st = i >> 1, kt = i & 1;
                kf[u] = *(const bf8*)(sK + lg * kKSub + (kt * 16 + l15) * kKRow + 8 * st);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kRA; ++u) {
                const int i = grp * kRA + u, st = i >> 1, kt = i & 1;
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u], qf[st], sc[kt], 0, 0, 0);
            }
        }
        // mask (only on blocks that touch the sequence end or the window edge) + online softmax over this
        // lane's 8 keys; the query's other 24 keys live in lanes ^16, ^32, ^48
        const uint32_t k_first = kb * 32u, k_last = k_first + 31u;
        const bool interior = k_last < L && (!window || ((q0 + 15u < k_first + window) && (k_last < q0 + window)));
        float mloc = -INFINITY;
        if (interior) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, sc[kt][r]);
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t key = k_first + (uint32_t)(kt * 16 + 4 * lg + r);
                    bool ok = key < L;
                    if (window) {
                        const uint32_t dist = key > qi ? key - qi : qi - key;
                        ok = ok && dist < window;
                    }
                    sc[kt][r] = ok ? sc[kt][r] : -INFINITY;
                    mloc = fmaxf(mloc, sc[kt][r]);
                }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        // Deferred rescale: the running max only moves when the block max exceeds it by more than
        // kRescaleThr; until then P = exp(s - m_run) <= e^8, exact in f32 and fine in bf16.
        const bool need = mloc > m_run + kRescaleThr || (m_run == -INFINITY && mloc != -INFINITY);
        if (__any(need)) {
            const float m_new = need ? mloc : m_run;
            const float a = need ? ((m_run == -INFINITY) ? 0.f : __expf(m_run - m_new)) : 1.f;
            l_run *= a;
#pragma unroll
            for (int d = 0; d < 16; ++d) o[d] *= a;
            m_run = m_new;
        }
        const float m_use = (m_run == -INFINITY) ? 0.f : m_run;   // no attendable key yet: every p = 0
        float lsum = 0.f;
        bf8 pf;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(sc[kt][r] - m_use);   // exp(-inf) = 0 for masked keys
                lsum += pv;
                pf[kt * 4 + r] = (bf16_t)pv;                    // slot (lg, 4kt + r) <-> key 16kt + 4lg + r
            }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        l_run += lsum;
        // slots 0..3 = keys 4lg + 0..3, slots 4..7 = keys 16 + 4lg + 0..3: one 16-B read per dim tile,
        // again kRA reads ahead of their MFMAs
#pragma unroll
        for (int grp = 0; grp < 16 / kRA; ++grp) {
            bf8 vf[kRA];
#pragma unroll
            for (int u = 0; u < kRA; ++u) vf[u] = *(const bf8*)(sV + lg * kVSub + ((grp * kRA + u) * 16 + l15) * 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kRA; ++u)
                o[grp * kRA + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u], pf, o[grp * kRA + u], 0, 0, 0);
        }
    }

    // O^T[d][q]: lane <-> query l15, register r of tile d <-> dim 16d + 4lg + r
    if (wave_live && qi < L) {
        const float invl = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* op = out + (size_t)(s0 + qi) * (heads * kHD) + head * kHD;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            bf4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(o[d][e] * invl);
            *(bf4*)(op + 16 * d + 4 * lg) = w;
        }
    }
}

// ---- attention, second generation: 64-key blocks, LDS-DMA double buffer ---------------------------------
// Same math and wave layout as attention_kernel (wave = 16 queries of one q-head; S^T = K Q^T, O^T += V^T P^T), but
// the K / V tiles never pass through registers: each 64-key block (K and V rows as they lie in the qkv buffer, 64 x 256
// each = 32 KiB) is fetched by `global_load_lds_dwordx4` into the buffer the previous block is not using, ONE barrier
// per 64 keys (attention_kernel: two per 32, plus 6 ds_write per thread), the softmax's cross-lane steps use
// v_permlane{16,32}_swap instead of ds_bpermute (which queues behind the fragment reads), and the PV product's V^T
// fragments come from the ROW-major V image through `ds_read_b64_tr_b16` (hardware transpose: no V^T buffer, no
// v_transpose launch on this path).
// LDS images are unpadded; the bank swizzle is applied to the DMA's SOURCE address (the destination is lane-linear):
//   K [key row 0..63][32 chunks of 8 dims]   chunk c of row r lives at c ^ fK(r), fK(r) = 4 ((r >> 3) & 3) + (r & 3)
//   V [key row 0..63][32 chunks of 8 dims]   chunk c of row r lives at c ^ fV(r), fV(r) = 2 (r & 3) + 8 ((r >> 3) & 1)
// The S^T tile (t, kt) takes its 16 key rows in the order row(i) = 32 t + 8 (i >> 2) + 4 kt + (i & 3), so that lane
// group lg's C registers of tiles kt = 0, 1 hold the 8 CONSECUTIVE keys 32 t + 8 lg + 0..7 = k-indices 8 lg + 0..7 of
// the PV product, and fK(row(i)) = i.  A transposed read serves, per 16-lane group, 4 key rows x 16 dims: lane 4q + p
// supplies the address of (key 8 lg + 4 h + q, dims 16 dt + 4 p ..+3) and lane i receives dim 16 dt + i of the 4 keys;
// h = 0, 1 give the 8 k-indices.  fV makes the 8 rows a 32-lane half touches land in 8 distinct 32-byte bank slots.
__device__ __forceinline__ float xor16_max(float v) {
    typedef unsigned pu2 __attribute__((ext_vector_type(2)));
    const pu2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    typedef unsigned pu2 __attribute__((ext_vector_type(2)));
    const pu2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float xor16_sum(float v) {
    typedef unsigned pu2 __attribute__((ext_vector_type(2)));
    const pu2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    typedef unsigned pu2 __attribute__((ext_vector_type(2)));
    const pu2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

constexpr int kABlk = 64;                    // keys per block
constexpr int kABuf = kABlk * kHD;           // elements of one K (or V^T) buffer: 32 KiB
constexpr size_t kAttDmaLds = (size_t)4 * kABuf * sizeof(bf16_t);   // [2] K + [2] V^T = 128 KiB

template <int WAVES, int G, int KRA_F>   // KRA_F: fragment reads in flight ahead of their MFMAs
__global__ __launch_bounds__(64 * WAVES) void attention_dma_kernel(const bf16_t* __restrict__ qkv,
                                                        const bf16_t* __restrict__ vt,
                                                        bf16_t* __restrict__ out,
                                                        const int32_t* __restrict__ blk,
                                                        const int32_t* __restrict__ seq_start,
                                                        const int32_t* __restrict__ seq_len,
                                                        const int32_t* __restrict__ vt_start, uint32_t vt_ld,
                                                        uint32_t heads, uint32_t kv_heads, uint32_t window,
                                                        const float* __restrict__ q_norm_w,
                                                        const float* __restrict__ cos_sin, float eps, float q_scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t asmem[];   // K[2][64 x 256] | V^T[2][256 x 64]
    constexpr int TQ = WAVES / G;                   // query tiles
    constexpr uint32_t kParts = 128 / (16 * TQ);    // workgroups per 128-query super-block of the blk list
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    // Workgroups go to the 8 XCDs round-robin by blockIdx.x; give each XCD one CONTIGUOUS run of the (sequence, query
    // block) list, so the workgroups that share a sequence's K / V^T share an L2 (blockIdx.x-order put them on 8
    // different XCDs: PMC showed 143 MB fetched per launch for 42 MB of q/k/v, every L2 streaming every sequence).
    const uint32_t nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const uint32_t wg = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const uint32_t sblk = wg / kParts, part = wg % kParts;
    const uint32_t b = (uint32_t)blk[2 * sblk], sb = (uint32_t)blk[2 * sblk + 1];
    const uint32_t head = blockIdx.y * (uint32_t)G + (uint32_t)(wid % G);
    const uint32_t g = head / (heads / kv_heads);
    const uint32_t s0 = (uint32_t)seq_start[b], L = (uint32_t)seq_len[b];   // (vt / vt_start / vt_ld: unused here)
    const uint32_t ld = (heads + 2u * kv_heads) * kHD;
    const uint32_t koff = (heads + g) * kHD;
    const uint32_t qbase = sb * 128u + part * (16u * TQ);      // the workgroup's first query
    if (qbase >= L) return;                                    // (uniform: before any barrier)
    const uint32_t q0 = qbase + (uint32_t)(wid / G) * 16u;     // this wave's first query
    const bool wave_live = q0 < L;                          // waves past the sequence only help staging
    const uint32_t qi = q0 + (uint32_t)l15;                 // this lane's query

    // 64-key blocks that can hold an attendable key: workgroup range (staging) and this wave's own range
    const uint32_t nkb = (L + (uint32_t)kABlk - 1u) / (uint32_t)kABlk;
    uint32_t kb_lo = 0, kb_hi = nkb, wkb_lo = 0, wkb_hi = nkb;
    if (window) {
        const uint32_t glo = qbase, ghi = glo + 16u * TQ - 1u;     // workgroup's queries
        kb_lo = (glo + 1u > window) ? (glo + 1u - window) / (uint32_t)kABlk : 0u;
        kb_hi = (ghi + window - 1u) / (uint32_t)kABlk + 1u;
        if (kb_hi > nkb) kb_hi = nkb;
        const uint32_t qhi = q0 + 15u;                              // this wave's queries
        wkb_lo = (q0 + 1u > window) ? (q0 + 1u - window) / (uint32_t)kABlk : 0u;
        wkb_hi = (qhi + window - 1u) / (uint32_t)kABlk + 1u;
    }

    // LDS-DMA of one block: 32 K instructions + 32 V instructions of 2 key rows (512 B each), instruction i by wave
    // i % WAVES.  M0 = LDS byte address of the instruction's 1 KiB; lane l lands at byte 16 l.  Rows past the sequence
    // read its last key (finite; masked / multiplied by P = 0 later).
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)asmem;
    const char* const gK = (const char*)(qkv + (size_t)s0 * ld + koff);
    const uint32_t vrel = kv_heads * (uint32_t)kHD * 2u;          // byte distance from a token's k head to its v head
    auto dma = [&](const char* sbase, uint32_t voff, uint32_t lds_byte) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds_byte), "v"(voff), "s"(sbase) : "memory");
    };
    constexpr int NI = (32 + WAVES - 1) / WAVES;
    auto stage = [&](uint32_t kb, uint32_t p) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const uint32_t i = (uint32_t)wid + (uint32_t)(WAVES * j);
            if (32 % WAVES != 0 && i >= 32u) break;                 // wave-uniform
            const uint32_t r = 2u * i + (uint32_t)(lane >> 5);
            uint32_t key = kb * (uint32_t)kABlk + r;
            key = key < L ? key : L - 1u;
            const uint32_t kc = (uint32_t)(lane & 31) ^ (4u * ((r >> 3) & 3u) + (r & 3u));
            const uint32_t vc = (uint32_t)(lane & 31) ^ (2u * (r & 3u) + 8u * ((r >> 3) & 1u));
            dma(gK, (key * ld + kc * 8u) * 2u, lds0 + p * (uint32_t)(kABuf * 2) + i * 1024u);
            dma(gK, (key * ld + vc * 8u) * 2u + vrel, lds0 + (uint32_t)(2 * kABuf * 2) + p * (uint32_t)(kABuf * 2) + i * 1024u);
        }
    };
    if (kb_lo < kb_hi) stage(kb_lo, 0u);      // in flight under the Q prologue
#ifdef CQS_ATT2_NO_QNORM
    q_norm_w = nullptr;
#endif

    const uint32_t qclamp = qi < L ? qi : L - 1u;
    bf8 qf[8];
    load_q_fragments(qf, qkv + (size_t)(s0 + qclamp) * ld + head * kHD, lg, q_norm_w, cos_sin + (size_t)qclamp * 256u, eps, q_scale);
    // Pin the fragments here: with their loads still "pending" at the loop head (the q_norm_w == NULL path), hipcc
    // places its s_waitcnt vmcnt(n..0) inside the loop body, where it would drain the next block's DMA every iteration.
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(qf[s]));

    // fragment addresses (elements, buffer 0).  K: row(i = l15) of tile (t, kt) = 32 t + 4 kt + 8 (l15 >> 2) + (l15 & 3),
    // chunk (4 st + lg) ^ l15 = 16 (st >> 2) + 4 ((st & 3) ^ (l15 >> 2)) + (lg ^ (l15 & 3)): four lane-dependent bases.
    const bf16_t* kp[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        kp[s] = asmem + (uint32_t)(8 * (l15 >> 2) + (l15 & 3)) * (uint32_t)kHD +
                (uint32_t)(4 * (s ^ (l15 >> 2)) + (lg ^ (l15 & 3))) * 8u;
    // V (transposed reads): lane 16 lg + 4 q + p addresses key row 8 lg + q (+ 4 h + 32 t), chunk 2 dt + (p >> 1), half
    // (p & 1); the chunk's position is 2 (dt ^ xq) + (p >> 1) with xq = q + 4 (lg & 1): eight lane-dependent bases by dt & 7.
    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) tr4* lds_tr4;
    const int tq = (lane >> 2) & 3, tp = lane & 3, xq = tq + 4 * (lg & 1);
    uint32_t vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        vb[j] = lds0 + (uint32_t)(2 * kABuf * 2) + (uint32_t)(8 * lg + tq) * (uint32_t)(kHD * 2) +
                (uint32_t)(2 * (j ^ xq) + (tp >> 1)) * 16u + (uint32_t)(tp & 1) * 8u;

    f4v o[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) o[d] = (f4v)(0.f);
    float m_run = -INFINITY, l_run = 0.f;

#ifdef CQS_ATT2_NO_LOOP
    kb_hi = kb_lo + 1u;
#endif
    // One 64-key block out of buffer P (compile-time: the buffer offset rides in the ds_read immediates, no per-block
    // pointer arithmetic and no second set of address registers)
    auto block = [&](uint32_t kb, auto par_c) {
        constexpr uint32_t p = (uint32_t)decltype(par_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of block kb has landed
        __syncthreads();                                       // everyone's has; block kb - 1 is fully consumed
#ifndef CQS_ATT2_NO_DMA
        if (kb + 1u < kb_hi) stage(kb + 1u, p ^ 1u);
#endif
        if (!wave_live || kb < wkb_lo || kb >= wkb_hi) return;     // wave-uniform
#ifdef CQS_ATT2_NO_COMPUTE
        if (kb != kb_lo) return;
#endif
        constexpr uint32_t pofs = p * (uint32_t)kABuf;
        const uint32_t kb_first = kb * (uint32_t)kABlk;
        using FencedRA = std::integral_constant<int, KRA_F>;
        // S^T tiles kt = 0, 1 of half t: 16 MFMAs, fragment reads KRA ahead.  `fenced`: keep each read group ahead of its
        // MFMA group (left alone, hipcc emits read -> wait -> MFMA pairs and every MFMA eats a full LDS round trip);
        // unfenced, the caller lays the schedule down with sched_group_barrier.
        auto s_phase = [&](int t, f4v (&sc)[2], auto ra_c) {
            constexpr int RA = decltype(ra_c)::value;      // > 0: fenced groups of RA reads; < 0: unfenced, -RA
            constexpr bool fenced = RA > 0;
            constexpr int KRA = RA > 0 ? RA : -RA;
            sc[0] = (f4v)(0.f);
            sc[1] = (f4v)(0.f);
#pragma unroll
            for (int grp = 0; grp < 16 / KRA; ++grp) {
                bf8 kf[KRA];
#pragma unroll
                for (int u = 0; u < KRA; ++u) {
                    const int i = grp * KRA + u, st = i >> 1, kt = i & 1;
#if defined(CQS_ATT2_NO_LDSREAD)
                    kf[u] = qf[(st + kt) & 7];
#else
                    kf[u] = *(const bf8*)(kp[st & 3] + pofs + (uint32_t)((32 * t + 4 * kt) * kHD + (st >> 2) * 128));
#endif
                }
                if (fenced) __builtin_amdgcn_sched_barrier(0);
#ifndef CQS_ATT2_NO_S
#pragma unroll
                for (int u = 0; u < KRA; ++u) {
                    const int i = grp * KRA + u, st = i >> 1, kt = i & 1;
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u], qf[st], sc[kt], 0, 0, 0);
                }
#else
#pragma unroll
                for (int u = 0; u < KRA; ++u) sc[u & 1][0] += (float)kf[u][0];
#endif
            }
        };
        // O^T += V^T P^T over half t: 16 dim tiles, one 16-byte fragment each
        auto pv_phase = [&](int t, const bf8& pf, auto ra_c) {
            constexpr int RA = decltype(ra_c)::value;
            constexpr bool fenced = RA > 0;
            constexpr int KRA = RA > 0 ? RA : -RA;
#pragma unroll
            for (int grp = 0; grp < 16 / KRA; ++grp) {
                bf8 vf[KRA];
#pragma unroll
#if defined(CQS_ATT2_NO_LDSREAD)
                for (int u = 0; u < KRA; ++u) vf[u] = qf[(grp + u) & 7];
#else
                for (int u = 0; u < KRA; ++u) {
                    const int dt = grp * KRA + u;
                    const uint32_t ad = vb[dt & 7] + pofs * 2u + (uint32_t)(t * 32 * kHD * 2 + (dt >> 3) * 256);
                    const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4)(uintptr_t)ad);
                    const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4)(uintptr_t)(ad + (uint32_t)(4 * kHD * 2)));
                    vf[u] = __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#endif
                if (fenced) __builtin_amdgcn_sched_barrier(0);
#ifndef CQS_ATT2_NO_PV
#pragma unroll
                for (int u = 0; u < KRA; ++u)
                    o[grp * KRA + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u], pf, o[grp * KRA + u], 0, 0, 0);
#else
#pragma unroll
                for (int u = 0; u < KRA; ++u) o[grp * KRA + u][0] += (float)vf[u][0] * (float)pf[0];
#endif
            }
        };
        // Online softmax of one half (this lane: 8 keys of query qi; the other 24 live in lanes ^16, ^32, ^48), branch-free
        // up to the (rare) rescale: sc[kt][r] = score of key k_first + 8 lg + 4 kt + r; returns P as the PV product's B
        // fragment (k-index 8 lg + 4 kt + r).  Deferred rescale: the running max only moves when the half's max exceeds it
        // by more than kRescaleThr; until then P = exp(s - m_run) <= e^8, exact in f32 and fine in bf16.
        constexpr float kLog2e = 1.4426950408889634f;
        auto softmax = [&](const f4v (&sc)[2], float mloc, bf8& pf, float& m_old, float& lsum) -> bool {
            // (branch-free on purpose: one basic block with the MFMAs it is interleaved with; the caller applies the
            // rescale of the lanes that return true - `need` - in a separate, rarely taken block)
            mloc = xor32_max(xor16_max(mloc));
            const bool need = (mloc > m_run + kRescaleThr) | ((m_run == -INFINITY) & (mloc != -INFINITY));   // (no short-circuit: no branch)
            m_old = m_run;
            m_run = need ? mloc : m_run;
            const float mb = (m_run == -INFINITY) ? 0.f : m_run * kLog2e;   // no attendable key yet: every p = 0
            lsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#ifdef CQS_ATT2_NO_EXP
                    const float pv = sc[kt][r] - mb;
#else
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][r], kLog2e, -mb));   // exp(-inf) = 0 for masked keys
#endif
                    lsum += pv;
                    pf[kt * 4 + r] = (bf16_t)pv;
                }
            return need;
        };
        // lanes with `need` moved their running max from m_old to m_run: O and the row sum shrink by exp(m_old - m_run)
        // (0 when there was no max yet: exp2(-inf))
        auto rescale = [&](bool need, float m_old) {
            const float a = need ? __builtin_amdgcn_exp2f((m_old - m_run) * kLog2e) : 1.f;
            l_run *= a;
#pragma unroll
            for (int d = 0; d < 16; ++d) o[d] *= a;
        };
        auto max8 = [&](const f4v (&sc)[2]) {
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[kt][r]);
            return m;
        };

        // Edge blocks (sequence end or a window edge inside them) mask key by key; a half without any attendable key
        // just yields P = 0.  (Measured and dropped: half 1's S^T MFMAs issued between half 0's softmax VALU, half 0's
        // PV MFMAs between half 1's softmax - sched_group_barrier interleave, 2 reads ahead to stay in 168 VGPRs:
        // 44.9 us vs 42.5.  The loop is co-bound: per 32 keys the CU's 12 waves need 1536 clk of MFMA per SIMD, 1536 clk
        // of LDS fragment reads and ~1000 clk of VALU, and removing any ONE of them leaves the time unchanged.)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t k_first = kb_first + 32u * (uint32_t)t, k_last = k_first + 31u;
            if (k_first >= L) break;                               // wave-uniform: the half lies past the sequence
            if (window && (k_last + window <= q0 || k_first >= q0 + 15u + window)) continue;   // no query of the wave sees it
            f4v sc[2];
            s_phase(t, sc, FencedRA{});
            // sc[kt][r] = score of key k_first + 8 lg + 4 kt + r for query qi
            const bool interior = k_last < L && (!window || ((q0 + 15u < k_first + window) && (k_last < q0 + window)));
            if (!interior) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t key = k_first + (uint32_t)(8 * lg + 4 * kt + r);
                        bool ok = key < L;
                        if (window) {
                            const uint32_t dist = key > qi ? key - qi : qi - key;
                            ok = ok && dist < window;
                        }
                        sc[kt][r] = ok ? sc[kt][r] : -INFINITY;
                    }
            }
            bf8 pf;
            float m_old, lsum;
            const bool need = softmax(sc, max8(sc), pf, m_old, lsum);
            if (__any(need)) rescale(need, m_old);
            l_run += lsum;              // per-lane partial of the row sum; reduced over the lane groups at the end
            pv_phase(t, pf, FencedRA{});
        }
    };
    for (uint32_t kb = kb_lo; kb < kb_hi; kb += 2u) {
        block(kb, std::integral_constant<int, 0>{});
        if (kb + 1u < kb_hi) block(kb + 1u, std::integral_constant<int, 1>{});
    }

    // O^T[d][q]: lane <-> query l15, register r of tile d <-> dim 16d + 4lg + r.  Through the wave's own 8 KiB of LDS
    // (the K buffers are dead after the barrier) so that the global stores are whole 512-byte head rows, 16 B per lane:
    // direct 8-byte stores (16 rows x 32 B per instruction) cost 7.5 us of this kernel's 48.
    l_run = xor32_sum(xor16_sum(l_run));
    __syncthreads();
    {
        const float invl = l_run > 0.f ? 1.0f / l_run : 0.f;
        // row q (512 B = 32 chunks of 8 dims): dims 16d + 4lg + 0..3 = half (lg & 1) of chunk 2d + (lg >> 1), stored at
        // chunk position c ^ (q & 7) ... 8-byte writes: lanes l15 = 0..15 hit distinct rows (stride 512 B = same bank
        // group) -> the XOR spreads them over 8 of the 16 slots; 2-way on a write costs nothing extra.
        bf16_t* const so = asmem + (uint32_t)wid * (16u * kHD);
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            bf4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(o[d][e] * invl);
            const uint32_t c = (uint32_t)(2 * d + (lg >> 1)) ^ (uint32_t)(l15 & 7) ^ (uint32_t)((l15 >> 3) << 3);
            *(bf4*)(so + (uint32_t)l15 * kHD + c * 8u + (uint32_t)(lg & 1) * 4u) = w;
        }
        // (same wave wrote and reads: no barrier, the compiler's lgkmcnt wait orders the LDS accesses)
        const uint32_t rr = (uint32_t)(lane >> 5), cc = (uint32_t)(lane & 31);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t q = 2u * (uint32_t)i + rr;                      // query row of the wave's tile
            const uint32_t c = cc ^ (q & 7u) ^ ((q >> 3) << 3);
            const u4 v = *(const u4*)(so + q * kHD + c * 8u);
            if (wave_live && q0 + q < L)
                *(u4*)(out + (size_t)(s0 + q0 + q) * (heads * kHD) + head * kHD + cc * 8u) = v;
        }
    }
}

// ---- GEMM (first generation; gemm_kernels.hip holds the 256-row ping-pong kernel that takes the full rounds) ----
// C[M,N] = A[M,K] W[N,K]^T, 128x128x64 tiles, 4 waves (2x2) of 64x64, 32x32x16 bf16 MFMA
// Measured alternatives at M=16384 (tools/gemm_bench.py), all 620-730 TF like this one: register-staged
// operands (ds_write_b128), a 256x128 tile with 4 waves of 128x64 and a 3-slot DMA ring (1 wave/SIMD: the
// ~100-cycle DMA issue cannot overlap the wave's own MFMAs: 1.5x slower), the same tile with 8 waves and
// staggered DMA issue (equal), BK = 32 with three workgroups per CU (620-670 TF: twice the barriers cost more than
// the third wave per SIMD gives back).  PMC on this kernel: waves issue 35 % of their cycles, are issue-stalled 41 %
// (mostly behind the other wave's MFMA) and parked at a waitcnt / barrier 24 %; the MFMA pipe is busy 36 %.
// The remaining gap to the matrix-core peak is per-K-step latency exposure
// (barrier + first fragment reads); closing it needs the phase-interleaved 256x256 schedule.
// LDS tile [128 rows][64 k] bf16, 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7): the 16
// rows one ds_read_b128 lane group touches then hit 16 distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ uint32_t swz(uint32_t row, uint32_t chunk) { return row * 64u + ((chunk ^ ((row >> 1) & 7u)) * 8u); }

__device__ __forceinline__ float gelu_erf(float x) {      // as p8_gelu_erf (gemm_kernels.hip): Abramowitz-Stegun 7.1.26
    const float z = __builtin_fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + __builtin_copysignf(e, x));
}

__device__ __forceinline__ float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): one v_exp + one v_rcp
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = 2.0f * k0 * (x + k1 * x * x * x);
    return x * __frcp_rn(1.0f + __expf(-u2));
}

template <int OUT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                        void* __restrict__ Cv, uint32_t M, uint32_t N, uint32_t K,
                                                        uint32_t ldc, const float* __restrict__ bias /*nullable; not GEGLU*/) {
    // ONE shared array (a second __shared__ object beside an LDS-DMA staging array can make hipcc
    // drain vmcnt before every ds_read): [buf][A|B][128 rows][64 k]
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 2 * 128 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;

    // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch); give each XCD
    // a contiguous run of tiles (n fastest) so the tiles sharing an A panel meet in one L2.
    const uint32_t nt = N / 128u, mt = (M + 127u) / 128u, total = nt * mt;
    const uint32_t bid = blockIdx.x, xcd = bid % 8u, q = total / 8u, r = total % 8u;
    const uint32_t tile = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + bid / 8u;
    const uint32_t m0 = (tile / nt) * 128u, n0 = (tile % nt) * 128u;

    // LDS-DMA staging (global_load_lds, 16 B per lane): one wave instruction fills 1 KiB = 8 tile
    // rows, lane l -> row (l >> 3), physical 16-B chunk (l & 7).  The swizzle therefore goes on the
    // SOURCE: the lane fetches logical chunk (l & 7) ^ ((row >> 1) & 7) of its row.  Each wave stages
    // 32 rows of A and 32 rows of B per K-step (4 + 4 instructions).
    const bf16_t* ga[4];
    const bf16_t* gb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t row = (uint32_t)(wid * 32 + u * 8 + (lane >> 3));
        const uint32_t c = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u);
        uint32_t ar = m0 + row;
        ar = ar < M ? ar : M - 1u;
        ga[u] = A + (size_t)ar * K + c * 8u;
        gb[u] = W + (size_t)(n0 + row) * K + c * 8u;
    }
    auto stage = [&](uint32_t kt, int buf) {
        bf16_t* dA = smem + (size_t)buf * (2 * 128 * 64) + (size_t)(wid * 32) * 64;
        bf16_t* dB = dA + 128 * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[u] + (size_t)kt * 64u),
                                             (__attribute__((address_space(3))) void*)(dA + u * 8 * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[u] + (size_t)kt * 64u),
                                             (__attribute__((address_space(3))) void*)(dB + u * 8 * 64), 16, 0, 0);
        }
    };
    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const uint32_t nk = K / 64u;
    stage(0, 0);
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const int buf = (int)(kt & 1u);
        // one barrier per K-step: it drains this wave's DMA (tile kt has landed for everyone) and
        // proves every wave is done reading the other buffer, which the next stage overwrites
        __syncthreads();
        if (kt + 1u < nk) stage(kt + 1u, buf ^ 1);   // flies under this tile's MFMAs
        const bf16_t* sA = smem + (size_t)buf * (2 * 128 * 64);
        const bf16_t* sB = sA + 128 * 64;
        // fragments of k-step ks+1 are read while the MFMAs of k-step ks run (two register sets)
        bf8 af[2][2], bfr[2][2];
        auto read_frags = [&](int ks, int set) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t row = (uint32_t)(wm * 64 + i * 32 + l31);
                af[set][i] = *(const bf8*)(sA + swz(row, (uint32_t)(2 * ks + lh)));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t row = (uint32_t)(wn * 64 + j * 32 + l31);
                bfr[set][j] = *(const bf8*)(sB + swz(row, (uint32_t)(2 * ks + lh)));
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < 4) read_frags(ks + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C tile element (row = (e&3) + 8(e>>2) + 4lh, col = l31) of acc[i][j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t row = m0 + (uint32_t)(wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh);
            if (row >= M) continue;
            if (OUT == GEMM_OUT_GEGLU) {
                // this wave's 64 columns = 32 gate channels (j = 0) + the same 32 channels' up (j = 1)
                const uint32_t ch = (n0 + (uint32_t)(wn * 64)) / 2u + (uint32_t)l31;
                const float v = gelu_tanh(acc[i][0][e]) * acc[i][1][e];
                ((bf16_t*)Cv)[(size_t)row * ldc + ch] = (bf16_t)v;
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t col = n0 + (uint32_t)(wn * 64 + j * 32 + l31);
                    float v = acc[i][j][e];
                    if (bias) v += bias[col];
                    if (OUT == GEMM_OUT_BF16_GELU) v = gelu_erf(v);
                    if (OUT == GEMM_OUT_F32) ((float*)Cv)[(size_t)row * ldc + col] = v;
                    else ((bf16_t*)Cv)[(size_t)row * ldc + col] = (bf16_t)v;
                }
            }
        }
}

// ---- masked mean pool: block = (sequence, 256 hidden dims); 16 waves split the tokens, 4 loads in flight each ----
// (96 workgroups for 32 sequences x 768 dims: the parallelism has to come from inside the workgroup)
__global__ __launch_bounds__(1024) void mean_pool_kernel(const float* __restrict__ hidden,
                                                         const int32_t* __restrict__ seq_start,
                                                         const int32_t* __restrict__ seq_len,
                                                         bf16_t* __restrict__ pooled, uint32_t H) {
    __shared__ f4 part[16][64];
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u, col = blockIdx.y * 256u + lane * 4u;
    const uint32_t s0 = (uint32_t)seq_start[b], L = (uint32_t)seq_len[b];
    const uint32_t w = threadIdx.x >> 6;
    const float* base = hidden + (size_t)s0 * H + col;
    f4 acc = (f4)(0.f);
    uint32_t t = w;
    for (; t + 48u < L; t += 64u) {   // 4 independent loads per trip
        const f4 a0 = *(const f4*)(base + (size_t)t * H), a1 = *(const f4*)(base + (size_t)(t + 16u) * H);
        const f4 a2 = *(const f4*)(base + (size_t)(t + 32u) * H), a3 = *(const f4*)(base + (size_t)(t + 48u) * H);
        acc += (a0 + a1) + (a2 + a3);
    }
    for (; t < L; t += 16u) acc += *(const f4*)(base + (size_t)t * H);
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        f4 sum = part[0][lane];
#pragma unroll
        for (int i = 1; i < 16; ++i) sum += part[i][lane];
        const float inv = L ? 1.0f / (float)L : 0.f;   // zero mask -> zero vector (src/embedder/pooling.rs:113-119)
        bf4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(sum[i] * inv);
        *(bf4*)(pooled + (size_t)b * H + col) = o;
    }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (bf16_t)in[i];
}

// ---- launchers ---------------------------------------------------------------------------
hipError_t launch_embed_norm(const int32_t* tok, const bf16_t* emb, float scale, const float* w_in, float eps,
                             float* x, bf16_t* xn, uint32_t M, uint32_t H, hipStream_t st) {
    if (M == 0) return hipSuccess;
    const dim3 grid((M + 3u) / 4u), block(256);
    switch (H / 256u) {
        case 1: hipLaunchKernelGGL(embed_norm_kernel<1>, grid, block, 0, st, tok, emb, scale, w_in, eps, x, xn, M); break;
        case 2: hipLaunchKernelGGL(embed_norm_kernel<2>, grid, block, 0, st, tok, emb, scale, w_in, eps, x, xn, M); break;
        case 3: hipLaunchKernelGGL(embed_norm_kernel<3>, grid, block, 0, st, tok, emb, scale, w_in, eps, x, xn, M); break;
        case 4: hipLaunchKernelGGL(embed_norm_kernel<4>, grid, block, 0, st, tok, emb, scale, w_in, eps, x, xn, M); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int FINAL>
static hipError_t launch_add_norm_t(float* x, const bf16_t* y, const float* w_post, const float* w_next, float eps,
                                    bf16_t* xn, float* out, uint32_t M, uint32_t H, hipStream_t st) {
    const dim3 grid((M + 3u) / 4u), block(256);
    switch (H / 256u) {
        case 1: hipLaunchKernelGGL((add_norm_kernel<1, FINAL>), grid, block, 0, st, x, y, w_post, w_next, eps, xn, out, M); break;
        case 2: hipLaunchKernelGGL((add_norm_kernel<2, FINAL>), grid, block, 0, st, x, y, w_post, w_next, eps, xn, out, M); break;
        case 3: hipLaunchKernelGGL((add_norm_kernel<3, FINAL>), grid, block, 0, st, x, y, w_post, w_next, eps, xn, out, M); break;
        case 4: hipLaunchKernelGGL((add_norm_kernel<4, FINAL>), grid, block, 0, st, x, y, w_post, w_next, eps, xn, out, M); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_add_norm(float* x, const bf16_t* y, const float* w_post, const float* w_next, float eps,
                           bf16_t* xn, float* out, int final, uint32_t M, uint32_t H, hipStream_t st) {
    if (M == 0) return hipSuccess;
    return final ? launch_add_norm_t<1>(x, y, w_post, w_next, eps, xn, out, M, H, st)
                 : launch_add_norm_t<0>(x, y, w_post, w_next, eps, xn, out, M, H, st);
}

// ---- few-rows GEMM: small batches (a query, a handful of chunks) ---------------------------------------------
// At M = 32 tokens the 128 x 128 kernel puts 6-18 workgroups on the chip and takes 12 us per projection (4 per layer:
// 70 % of a query's 1.6 ms).  Here ONE WAVE owns a 32 x 32 output tile (GeGLU: 32 x 64 = the gate and up halves of 32
// channels) and walks K by itself with operands straight from global memory / L2 (16 B per lane per 16-k step, 8
// steps in flight), no LDS, no barrier: N / 32 waves per 32 rows.  Same MFMA (32x32x16), same operand roles and the
// same K order as gemm_bf16_kernel, so the two kernels agree bit for bit and a chunk still embeds to the same bits
// alone or in a batch (test_padding_and_batch_invariance).

template <int OUT>
__global__ __launch_bounds__(64) void gemm_fewrows_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                          void* __restrict__ Cv, uint32_t M, uint32_t N, uint32_t K,
                                                          uint32_t ldc, const float* __restrict__ bias /*nullable; not GEGLU*/) {
    constexpr int NT = OUT == GEMM_OUT_GEGLU ? 2 : 1;          // 32-column tiles per wave
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    const uint32_t n0 = blockIdx.x * (uint32_t)(32 * NT), m0 = blockIdx.y * 32u;
    const uint32_t mr = m0 + (uint32_t)l31 < M ? m0 + (uint32_t)l31 : M - 1u;     // rows past M: any real row, never stored
    const bf16_t* ap = A + (size_t)mr * K + 8 * lh;
    const bf16_t* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = W + (size_t)(n0 + (uint32_t)(32 * j + l31)) * K + 8 * lh;
    f16v acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const uint32_t steps = K / 16u;                             // K % 64 == 0
    constexpr int U = 8;
    for (uint32_t s = 0; s < steps; s += U) {
        bf8 af[U], wf[NT][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t ss = s + (uint32_t)u < steps ? s + (uint32_t)u : steps - 1u;     // (tail: re-read, not accumulated)
            af[u] = *(const bf8*)(ap + (size_t)ss * 16u);
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j][u] = *(const bf8*)(wp[j] + (size_t)ss * 16u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s + (uint32_t)u < steps) {
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u], wf[j][u], acc[j], 0, 0, 0);
            }
    }
    // C tile element (row = (e & 3) + 8 (e >> 2) + 4 lh, col = l31), as in gemm_bf16_kernel
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const uint32_t row = m0 + (uint32_t)((e & 3) + 8 * (e >> 2) + 4 * lh);
        if (row >= M) continue;
        if (OUT == GEMM_OUT_GEGLU) {
            const float v = gelu_tanh(acc[0][e]) * acc[NT - 1][e];
            ((bf16_t*)Cv)[(size_t)row * ldc + n0 / 2u + (uint32_t)l31] = (bf16_t)v;
        } else {
            float v = acc[0][e];
            if (bias) v += bias[n0 + (uint32_t)l31];
            if (OUT == GEMM_OUT_BF16_GELU) v = gelu_erf(v);
            if (OUT == GEMM_OUT_F32) ((float*)Cv)[(size_t)row * ldc + n0 + (uint32_t)l31] = v;
            else ((bf16_t*)Cv)[(size_t)row * ldc + n0 + (uint32_t)l31] = (bf16_t)v;
        }
    }
}

static hipError_t launch_gemm_fewrows(const bf16_t* A, const bf16_t* W, const float* bias, void* C, uint32_t M, uint32_t N,
                                      uint32_t K, uint32_t ldc, GemmOut out, hipStream_t st) {
    if (N % 64u || K % 64u || (bias && out == GEMM_OUT_GEGLU)) return hipErrorInvalidValue;
    const dim3 fg(N / (out == GEMM_OUT_GEGLU ? 64u : 32u), (M + 31u) / 32u);
    switch (out) {
        case GEMM_OUT_BF16: hipLaunchKernelGGL(gemm_fewrows_kernel<GEMM_OUT_BF16>, fg, dim3(64), 0, st, A, W, C, M, N, K, ldc, bias); break;
        case GEMM_OUT_F32: hipLaunchKernelGGL(gemm_fewrows_kernel<GEMM_OUT_F32>, fg, dim3(64), 0, st, A, W, C, M, N, K, ldc, bias); break;
        case GEMM_OUT_GEGLU: hipLaunchKernelGGL(gemm_fewrows_kernel<GEMM_OUT_GEGLU>, fg, dim3(64), 0, st, A, W, C, M, N, K, ldc, bias); break;
        case GEMM_OUT_BF16_GELU: hipLaunchKernelGGL(gemm_fewrows_kernel<GEMM_OUT_BF16_GELU>, fg, dim3(64), 0, st, A, W, C, M, N, K, ldc, bias); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// One kernel for the whole [M, N] problem; tn = 0: the 128 x 128 kernel, 3..5: the 256 x (64 tn) ping-pong kernel.
static hipError_t launch_gemm_one(const bf16_t* A, const bf16_t* W, void* C, uint32_t M, uint32_t N, uint32_t K,
                                  uint32_t ldc, GemmOut out, int tn, hipStream_t st, const float* bias = nullptr) {
    if (tn) return launch_gemm_p8(A, W, C, M, N, K, ldc, out, tn, st, bias);
    static const uint32_t few_max = [] { const char* f = getenv("CQS_HIP_GEMM_FEWROWS"); return f ? (uint32_t)atoi(f) : 512u; }();
    if (M <= few_max && !getenv("CQS_HIP_GEMM_TILE"))           // small batch: one wave per 32 x 32 tile (bit-identical results)
        return launch_gemm_fewrows(A, W, bias, C, M, N, K, ldc, out, st);
    const dim3 grid((N / 128u) * ((M + 127u) / 128u)), block(256);
    switch (out) {
        case GEMM_OUT_BF16: hipLaunchKernelGGL(gemm_bf16_kernel<GEMM_OUT_BF16>, grid, block, 0, st, A, W, C, M, N, K, ldc, bias); break;
        case GEMM_OUT_F32: hipLaunchKernelGGL(gemm_bf16_kernel<GEMM_OUT_F32>, grid, block, 0, st, A, W, C, M, N, K, ldc, bias); break;
        case GEMM_OUT_GEGLU: hipLaunchKernelGGL(gemm_bf16_kernel<GEMM_OUT_GEGLU>, grid, block, 0, st, A, W, C, M, N, K, ldc, nullptr); break;
        case GEMM_OUT_BF16_GELU: hipLaunchKernelGGL(gemm_bf16_kernel<GEMM_OUT_BF16_GELU>, grid, block, 0, st, A, W, C, M, N, K, ldc, bias); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- skinny GEMM: a handful of rows (the pooled sentence vectors through the two Dense layers: M = sequences of
// the batch) against a big weight matrix.  The tiled kernels put 24 (or 6) workgroups on the chip for M = 32 and take
// 43 + 14 us; this one is a batch of GEMVs: one workgroup = one 16 x 16 output tile, its 4 waves split K (each streams
// its quarter of the 16 weight rows once, 16 B per lane, loads 8 k-steps deep), partial tiles summed through LDS.
template <int OUT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                         void* __restrict__ Cv, uint32_t M, uint32_t N, uint32_t K,
                                                         uint32_t ldc, uint32_t lda /*row stride of A, elements*/,
                                                         const float* __restrict__ bias /*nullable*/, int act /*1: tanh*/,
                                                         const int32_t* __restrict__ row_index /*nullable: A row of output row m*/) {
    __shared__ __attribute__((aligned(16))) float red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const uint32_t n0 = blockIdx.x * 16u, m0 = blockIdx.y * 16u;
    const uint32_t ksteps = K / 32u, per = (ksteps + 3u) / 4u;
    const uint32_t s_lo = (uint32_t)wid * per, s_hi = s_lo + per < ksteps ? s_lo + per : ksteps;
    const uint32_t mr = m0 + (uint32_t)l15 < M ? m0 + (uint32_t)l15 : M - 1u;    // rows past M: any real row, never stored
    const bf16_t* ap = A + (size_t)(row_index ? (uint32_t)row_index[mr] : mr) * lda + 8 * lg;
    const bf16_t* wp = W + (size_t)(n0 + (uint32_t)l15) * K + 8 * lg;
    f4 acc = (f4)(0.f);
    constexpr int U = 8;
    for (uint32_t s = s_lo; s < s_hi; s += U) {
        bf8 af[U], wf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t ss = s + (uint32_t)u < s_hi ? s + (uint32_t)u : s_hi - 1u;   // (tail: re-read, not accumulated)
            af[u] = *(const bf8*)(ap + (size_t)ss * 32u);
            wf[u] = *(const bf8*)(wp + (size_t)ss * 32u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s + (uint32_t)u < s_hi) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[u], acc, 0, 0, 0);
    }
    // acc[r] = C[m = m0 + l15][n = n0 + 4 lg + r] (weights as the A operand: a lane holds 4 consecutive columns)
    *(f4*)&red[wid][(l15 * 4 + lg) * 4] = acc;
    __syncthreads();
    if (wid == 0) {
        f4 v = *(const f4*)&red[0][(l15 * 4 + lg) * 4];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *(const f4*)&red[w][(l15 * 4 + lg) * 4];
        const uint32_t m = m0 + (uint32_t)l15;
        if (bias) v += *(const f4*)(bias + n0 + 4 * lg);
        if (act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
        }
        if (m < M) {
            if (OUT == GEMM_OUT_F32) *(f4*)((float*)Cv + (size_t)m * ldc + n0 + 4 * lg) = v;
            else {
                bf4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
                *(bf4*)((bf16_t*)Cv + (size_t)m * ldc + n0 + 4 * lg) = o;
            }
        }
    }
}

// Kernel plan of one [M, N, K] projection: n1 columns with tile kind tn1 (0 = the 128 x 128 kernel, 3..5 = 256 x 64 tn),
// the remaining N - n1 columns (if any) with tn2.
struct GemmPlan { uint32_t n1; int tn1, tn2; };
static GemmPlan plan_gemm(uint32_t M, uint32_t N, uint32_t K, GemmOut out) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
    }
    // Kernel choice: rounds x measured cost of one round of tiles (microseconds at K = 768 on an MI355X, launch to
    // launch).  A round of the 256-row kernel costs ~8 us of prologue + epilogue on top of its K-steps and rounds of one
    // launch do not overlap (one workgroup per CU), so big tiles only pay when their rounds are FULL:
    //   128 x 128: 9 (GeGLU 8.2) | 256 x 192: 24 | 256 x 256: 25.5 | 256 x 320: 32.5   (GeGLU epilogue: + 1.5)
    // and a problem whose tile count is not a multiple of the CU count is cut in two launches along N: the part that
    // makes whole rounds of big tiles + the rest (N = 2304 at 16 384 rows: 2048 columns = 2 rounds of 256 x 256, then
    // 256 columns = one round of 128 x 128: 62 us instead of 73).
    const float kscale = (float)K / 768.f;
    const float geglu = out == GEMM_OUT_GEGLU ? 1.f : 0.f;
    static const uint32_t cu_env = [] { const char* f = getenv("CQS_HIP_GEMM_CUS"); const int v = f ? atoi(f) : 0; return v > 0 ? (uint32_t)v : 0u; }();
    const uint32_t cu = cu_env ? cu_env : (uint32_t)n_cu;      // (experiment hook: plan for part of the chip; read once)
    const bool fits = (uint64_t)M * K < (1ull << 31) && (uint64_t)N * K < (1ull << 31);
    const float cost[6] = {0.f, 0.f, 0.f, 24.f, 25.5f, 32.5f};
    auto one = [&](uint32_t n, int t) -> float {              // cost of n columns with one kernel; < 0: not applicable
        if (t == 0) return (float)(((n / 128u) * ((M + 127u) / 128u) + cu - 1u) / cu) * (3.f + (6.f - 0.8f * geglu) * kscale);
        if (!fits || n % (64u * (uint32_t)t)) return -1.f;
        return (float)(((n / (64u * (uint32_t)t)) * ((M + 255u) / 256u) + cu - 1u) / cu) * (8.f + 1.5f * geglu + (cost[t] - 8.f) * kscale);
    };
    auto best_one = [&](uint32_t n, int& t_out) -> float {
        float best = one(n, 0);
        t_out = 0;
        for (int t = 3; t <= 5; ++t) { const float c = one(n, t); if (c >= 0.f && c < best) { best = c; t_out = t; } }
        return best;
    };
    int tn1 = 0, tn2 = 0;
    uint32_t n1 = N;
    float best = best_one(N, tn1);
    const uint32_t mt = (M + 255u) / 256u;
    uint32_t g = mt, h = cu;
    while (h) { const uint32_t r = g % h; g = h; h = r; }      // g = gcd(mt, cu)
    for (int t = 3; t <= 5 && fits; ++t) {
        const uint32_t step = (cu / g) * 64u * (uint32_t)t;     // columns that make whole rounds of 256 x 64 t tiles
        for (uint32_t c1 = step; c1 < N; c1 += step) {
            if ((N - c1) % 128u) continue;
            int t2 = 0;
            const float c = one(c1, t) + best_one(N - c1, t2) + 2.f;   // + one kernel boundary
            if (c < best) { best = c; n1 = c1; tn1 = t; tn2 = t2; }
        }
    }
    if (const char* f = getenv("CQS_HIP_GEMM_TILE")) {  // test hook: "small" / "pp:<tn>" force one kernel (read per call: tests flip it)
        n1 = N;
        if (f[0] == 's') tn1 = 0;
        else if (f[0] == 'p') {
            const int t = f[2] == ':' ? atoi(f + 3) : 4;
            tn1 = (t >= 3 && t <= 5 && fits && N % (64u * (uint32_t)t) == 0) ? t : 0;
        }
    }
    return {n1, tn1, tn2};
}

int gemm_qkv_rope_tile(uint32_t M, uint32_t hidden, uint32_t heads, uint32_t kv_heads, uint32_t head_dim) {
    if (head_dim != 256u || kv_heads == 0u || hidden % 64u) return 0;
    const uint32_t N = (heads + 2u * kv_heads) * head_dim;
    const GemmPlan p = plan_gemm(M, N, hidden, GEMM_OUT_BF16);
    if (p.n1 != N) return 0;                              // the plain projection would be one launch of that tile too
    if (p.tn1 == 5 && heads == 3u * kv_heads) return 5;
    if (p.tn1 == 4) return 4;
    return 0;
}

hipError_t launch_gemm_bf16(const bf16_t* A, const bf16_t* W, void* C, uint32_t M, uint32_t N, uint32_t K,
                            uint32_t ldc, GemmOut out, hipStream_t st, const float* bias, const bf16_t* W_geglu4) {
    if (M == 0) return hipSuccess;
    if (N % 128u || K % 64u || (bias && out == GEMM_OUT_GEGLU)) return hipErrorInvalidValue;
    const GemmPlan pl = plan_gemm(M, N, K, out);
    const uint32_t n1 = pl.n1;
    const int tn1 = pl.tn1, tn2 = pl.tn2;
    if (out == GEMM_OUT_GEGLU && W_geglu4) {
        // parts the 256-row kernel takes read the per-4 interleave and pair gate / up in registers; a part on the 128 x 128 /
        // few-rows kernels keeps the per-32 order (cuts are multiples of 64 rows: the same channels on either side in both)
        const size_t coff = n1 / 2u;
        if (n1 == N) return launch_gemm_one(A, tn1 ? W_geglu4 : W, C, M, N, K, ldc, tn1 ? GEMM_OUT_GEGLU4 : GEMM_OUT_GEGLU, tn1, st, nullptr);
        if (n1 % 64u) return hipErrorInvalidValue;
        static const bool no_dual4 = getenv("CQS_HIP_GEMM_NO_DUAL") != nullptr;
        if (tn1 >= 3 && tn2 >= 3 && tn1 != tn2 && !no_dual4) {
            const hipError_t d = launch_gemm_p8_dual(A, W_geglu4, C, n1, tn1, W_geglu4 + (size_t)n1 * K, (bf16_t*)C + coff, N - n1, tn2, M, K, ldc, GEMM_OUT_GEGLU4, st);
            if (d != hipErrorNotSupported) return d;
        }
        hipError_t e = launch_gemm_one(A, tn1 ? W_geglu4 : W, C, M, n1, K, ldc, tn1 ? GEMM_OUT_GEGLU4 : GEMM_OUT_GEGLU, tn1, st, nullptr);
        if (e != hipSuccess) return e;
        return launch_gemm_one(A, (tn2 ? W_geglu4 : W) + (size_t)n1 * K, (bf16_t*)C + coff, M, N - n1, K, ldc, tn2 ? GEMM_OUT_GEGLU4 : GEMM_OUT_GEGLU, tn2, st, nullptr);
    }
    if (n1 == N) return launch_gemm_one(A, W, C, M, N, K, ldc, out, tn1, st, bias);
    const size_t coff = out == GEMM_OUT_GEGLU ? n1 / 2u : n1;    // output columns of the first part
    void* c2 = out == GEMM_OUT_F32 ? (void*)((float*)C + coff) : (void*)((bf16_t*)C + coff);
    static const bool no_dual = getenv("CQS_HIP_GEMM_NO_DUAL") != nullptr;                       // (read once)
    if (tn1 >= 3 && tn2 >= 3 && tn1 != tn2 && !bias && out != GEMM_OUT_BF16_GELU && !no_dual) {   // both parts in one launch
        const hipError_t d = launch_gemm_p8_dual(A, W, C, n1, tn1, W + (size_t)n1 * K, c2, N - n1, tn2, M, K, ldc, out, st);
        if (d != hipErrorNotSupported) return d;
    }
    hipError_t e = launch_gemm_one(A, W, C, M, n1, K, ldc, out, tn1, st, bias);
    if (e != hipSuccess) return e;
    return launch_gemm_one(A, W + (size_t)n1 * K, c2, M, N - n1, K, ldc, out, tn2, st, bias ? bias + n1 : nullptr);
}

hipError_t launch_gemm_bias(const bf16_t* A, const bf16_t* W, const float* bias, void* C, uint32_t M, uint32_t N,
                            uint32_t K, uint32_t ldc, GemmOut out, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (out == GEMM_OUT_GEGLU || K % 64u) return hipErrorInvalidValue;
    // small batches (a query, a rerank of a few dozen passages): one wave per 32 x 32 tile, no fixed cost of the 256-row
    // kernel's prologue / epilogue (13-50 us per projection at a few thousand tokens; measured crossover below)
    // Crossover measured on whole forwards (tools/bert_fewrows_sweep.py): few-rows wins up to ~900 tokens for BERT-base
    // (hidden 768), ~600 for BERT-large (1024), ~2000 for MiniLM (384) - for ALL of a layer's projections, the K = 4 x
    // hidden one included - i.e. tokens x hidden <~ 640 Ki; min(N, K) is the hidden size of every BERT projection.
    // up to 64 rows (a SPLADE query, one short passage): the search-time kernels - K split over a workgroup's waves, 96-384
    // workgroups - instead of 24-96 lone waves walking all of K (BERT-base FFN2 at 16 tokens: 25 us -> 3 us)
    if (M <= 64u) {
        const char* sr = getenv("CQS_HIP_GEMM_SMALL_ROWS");       // read per call: a test flips it inside one process
        if (!(sr && sr[0] == '0')) {
            const hipError_t e = launch_gemm_small_rows(A, W, bias, C, M, N, K, ldc, out, st);
            if (e != hipErrorNotSupported) return e;
        }
    }
    static const uint64_t few_mh = [] { const char* f = getenv("CQS_HIP_GEMM_BIAS_FEWROWS_MH"); return f ? (uint64_t)atoll(f) : 640ull * 1024ull; }();
    if ((uint64_t)M * (N < K ? N : K) <= few_mh && N % 64u == 0) return launch_gemm_fewrows(A, W, bias, C, M, N, K, ldc, out, st);
    // just above the few-rows range the 128 x 128 kernel still beats a mostly empty round of 256-row tiles (measured:
    // 1024 tokens of BERT-base 1.72 -> 1.62 ms, BERT-large 4.6 -> 4.15 ms; from ~2k tokens on the 256-row kernel wins)
    if (N % 128u == 0 && M <= 1536u) return launch_gemm_bf16(A, W, C, M, N, K, ldc, out, st, bias);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
    }
    // (N not a multiple of 128) rounds x cost of a round (launch_gemm_bf16's table), over the tile widths that divide N
    const float cost[6] = {0.f, 0.f, 0.f, 24.f, 25.5f, 32.5f};
    int best_t = 0;
    float best = 0.f;
    for (int t = 3; t <= 5; ++t) {
        if (N % (64u * (uint32_t)t)) continue;
        const uint32_t tiles = (N / (64u * (uint32_t)t)) * ((M + 255u) / 256u);
        const float c = (float)((tiles + (uint32_t)n_cu - 1u) / (uint32_t)n_cu) * (8.f + (cost[t] - 8.f) * (float)K / 768.f);
        if (!best_t || c < best) { best = c; best_t = t; }
    }
    if (!best_t) return hipErrorInvalidValue;
    return launch_gemm_p8(A, W, C, M, N, K, ldc, out, best_t, st, bias);
}

// The Dense head's GEMMs (M = sequences of the batch).  Not chosen by launch_gemm_bf16 itself: its K split sums in a
// different order than the tiled kernels, and a token's activations must not depend on how many tokens share its
// batch (tests/test_embed_gpu.py::test_padding_and_batch_invariance) - the head, applied once per sequence, always
// takes this path up to 256 sequences, the tiled kernels beyond.
hipError_t launch_gemm_skinny(const bf16_t* A, const bf16_t* W, void* C, uint32_t M, uint32_t N, uint32_t K,
                              uint32_t ldc, GemmOut out, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (N % 16u || K % 32u || (out != GEMM_OUT_BF16 && out != GEMM_OUT_F32)) return hipErrorInvalidValue;
    if (M > 256u) return launch_gemm_bf16(A, W, C, M, N, K, ldc, out, st);
    return launch_gemm_rows(A, K, W, nullptr, 0, C, M, N, K, ldc, out, st, nullptr);
}

// The same kernel on strided rows with a bias and an optional tanh (the BERT pooler reads every sequence's first
// token out of the packed hidden states: lda = its stride; any M).
hipError_t launch_gemm_rows(const bf16_t* A, uint32_t lda, const bf16_t* W, const float* bias, int act_tanh, void* C,
                            uint32_t M, uint32_t N, uint32_t K, uint32_t ldc, GemmOut out, hipStream_t st,
                            const int32_t* row_index) {
    if (M == 0) return hipSuccess;
    if (N % 16u || K % 32u || (out != GEMM_OUT_BF16 && out != GEMM_OUT_F32)) return hipErrorInvalidValue;
    const dim3 grid(N / 16u, (M + 15u) / 16u);
    if (out == GEMM_OUT_F32)
        hipLaunchKernelGGL(gemm_skinny_kernel<GEMM_OUT_F32>, grid, dim3(256), 0, st, A, W, C, M, N, K, ldc, lda, bias, act_tanh, row_index);
    else
        hipLaunchKernelGGL(gemm_skinny_kernel<GEMM_OUT_BF16>, grid, dim3(256), 0, st, A, W, C, M, N, K, ldc, lda, bias, act_tanh, row_index);
    return hipGetLastError();
}


hipError_t launch_qk_norm_rope(bf16_t* qkv, const int32_t* pos, const float* wq, const float* wk,
                               const float* cos_sin, float eps, float q_scale, uint32_t M, uint32_t heads,
                               uint32_t kv_heads, int k_only, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (heads + kv_heads > (uint32_t)kMaxQkHeads) return hipErrorInvalidValue;
    hipLaunchKernelGGL(qk_norm_rope_kernel, dim3((M + 3u) / 4u), dim3(256), 0, st, qkv, pos, wq, wk, cos_sin, eps,
                       q_scale, M, heads, kv_heads, k_only ? heads : 0u);
    return hipGetLastError();
}

hipError_t launch_kv_prep(bf16_t* qkv, bf16_t* vt, const int32_t* pos, const float* wq, const float* wk,
                          const float* cos_sin, float eps, float q_scale, uint32_t M, uint32_t heads, uint32_t kv_heads,
                          const int32_t* blk, uint32_t nblk, const int32_t* seq_start, const int32_t* seq_len,
                          const int32_t* vt_start, uint32_t vt_ld, int with_vt, hipStream_t st) {
    if (M == 0 || nblk == 0) return hipSuccess;
    if (kv_heads > (uint32_t)kMaxQkHeads) return hipErrorInvalidValue;
    const uint32_t n_rope = (M + 3u) / 4u, n_vt = with_vt ? nblk * 2u * kv_heads * 4u : 0u;
    hipLaunchKernelGGL(kv_prep_kernel, dim3(n_rope + n_vt), dim3(256), 0, st, qkv, vt, pos, wq, wk, cos_sin, eps, q_scale, M,
                       heads, kv_heads, n_rope, blk, nblk, seq_start, seq_len, vt_start, vt_ld);
    return hipGetLastError();
}

hipError_t launch_v_transpose(const bf16_t* qkv, bf16_t* vt, const int32_t* blk, uint32_t nblk,
                              const int32_t* seq_start, const int32_t* seq_len, const int32_t* vt_start, uint32_t heads,
                              uint32_t kv_heads, uint32_t vt_ld, hipStream_t st) {
    if (nblk == 0) return hipSuccess;
    hipLaunchKernelGGL(v_transpose_kernel, dim3(nblk * 2u, kv_heads * 4u), dim3(256), 0, st, qkv, vt, blk, seq_start, seq_len,
                       vt_start, heads, kv_heads, vt_ld);
    return hipGetLastError();
}

// Which attention kernel a batch gets.  With several q-heads per kv head: head-sharing workgroups on the 64-key LDS-DMA
// kernel (row-major V, no V^T), from ~1.5k tokens on - rounds 1-2 kept one q-head per workgroup on the register-staged
// kernel (which reads V^T) below one head-sharing workgroup per CU ("more, thinner workgroups fill the chip better": +6 %
// against round 1's head-sharing kernel); against the LDS-DMA kernel that rule lost 2-7 % of the whole forward at 4-48
// ragged sequences (tools/embed_att_layout_sweep.sh, round 3).  CQS_HIP_ATT_KERNEL=reg keeps the register-staged kernel,
// CQS_HIP_ATT_LAYOUT=shared / per-head forces a layout (test hooks: every combination is checked against the oracle).
struct AttPlan { bool share, dma; };
static AttPlan att_plan(uint32_t nblk, uint32_t heads, uint32_t kv_heads) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
    }
    const uint32_t ratio = kv_heads ? heads / kv_heads : 0u;
    bool share = ratio > 1u && nblk * 2u * kv_heads >= 24u;      // (under ~1.5k tokens - one sequence - the thin workgroups are as fast or faster)
    (void)n_cu;
    if (const char* f = getenv("CQS_HIP_ATT_LAYOUT")) {
        if (f[0] == 's') share = ratio > 1u;
        else if (f[0] == 'p') share = false;
    }
    bool dma = share && ratio >= 2u && ratio <= 4u;
    if (const char* f = getenv("CQS_HIP_ATT_KERNEL")) dma = dma && f[0] != 'r';
    return {share, dma};
}
bool attention_reads_vt(uint32_t nblk, uint32_t heads, uint32_t kv_heads) { return !att_plan(nblk, heads, kv_heads).dma; }

hipError_t launch_attention(const bf16_t* qkv, const bf16_t* vt, bf16_t* out, const int32_t* blk, uint32_t nblk,
                            const int32_t* seq_start, const int32_t* seq_len, const int32_t* vt_start, uint32_t vt_ld,
                            uint32_t heads, uint32_t kv_heads, uint32_t window, const float* q_norm_w,
                            const float* cos_sin, float eps, float q_scale, hipStream_t st) {
    if (nblk == 0) return hipSuccess;
    if (kv_heads == 0 || heads % kv_heads) return hipErrorInvalidValue;
#ifndef CQS_ATT_TQ
#define CQS_ATT_TQ 4
#endif
    // One workgroup = CQS_ATT_TQ tiles of 16 queries x all q-heads of a kv head (EmbeddingGemma: 4 x 3 = 12
    // waves, 64 queries): the staged K / V^T tiles serve 192 query-heads instead of 128 (8 waves x 1 head),
    // and 32 x 512-token sequences make 256 workgroups = one per CU in one round instead of 384.
    // (4 q-heads per kv head: 2 tiles, or the 16 waves would be held to 128 VGPRs and spill)
#define CQS_ATT(GV, TQV)                                                                                           \
    hipLaunchKernelGGL((attention_kernel<TQV * GV, GV>), dim3(nblk * (128 / (16 * TQV)), heads / GV),             \
                       dim3(64 * TQV * GV), 0, st, qkv, vt, out, blk, seq_start, seq_len, vt_start, vt_ld, heads,   \
                       kv_heads, window, q_norm_w, cos_sin, eps, q_scale)
    const uint32_t ratio = heads / kv_heads;
    const AttPlan plan = att_plan(nblk, heads, kv_heads);
    const bool share = plan.share, use_dma = plan.dma;
    if (!share) { CQS_ATT(1, 8); return hipGetLastError(); }
#define CQS_ATT_DMA(GV, TQV, KRAV)                                                                                  \
    do {                                                                                                            \
        auto kern = attention_dma_kernel<TQV * GV, GV, KRAV>;                                                       \
        static std::atomic<uint64_t> attr_devices{0};                                                               \
        {                                                                                                           \
            const hipError_t e = set_max_dynamic_lds((const void*)kern, kAttDmaLds, attr_devices);                  \
            if (e != hipSuccess) return e;                                                                          \
        }                                                                                                           \
        hipLaunchKernelGGL(kern, dim3(nblk * (128 / (16 * TQV)), heads / GV), dim3(64 * TQV * GV), kAttDmaLds, st,  \
                           qkv, vt, out, blk, seq_start, seq_len, vt_start, vt_ld, heads, kv_heads, window,         \
                           q_norm_w, cos_sin, eps, q_scale);                                                        \
    } while (0)
    if (use_dma) {
        switch (ratio) {
            case 2: CQS_ATT_DMA(2, 4, 8); break;
            case 3: CQS_ATT_DMA(3, CQS_ATT_TQ, 4); break;
            case 4: CQS_ATT_DMA(4, 2, 8); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
#undef CQS_ATT_DMA
    switch (ratio) {
        case 2: CQS_ATT(2, 4); break;
        case 3: CQS_ATT(3, CQS_ATT_TQ); break;
        case 4: CQS_ATT(4, 2); break;
        default: return hipErrorInvalidValue;
    }
#undef CQS_ATT
    return hipGetLastError();
}

hipError_t launch_mean_pool(const float* hidden, const int32_t* seq_start, const int32_t* seq_len, bf16_t* pooled,
                            uint32_t B, uint32_t H, hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(mean_pool_kernel, dim3(B, H / 256u), dim3(1024), 0, st, hidden, seq_start, seq_len, pooled, H);
    return hipGetLastError();
}

hipError_t launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

}  // namespace cqs
