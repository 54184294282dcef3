// bert_kernels.hip — gfx950 kernels of the BERT-family forwards behind cqs's two auxiliary models:
// the SPLADE sparse encoder (BERT-base masked-LM, src/splade/mod.rs) and the cross-encoder reranker (MiniLM-L6,
// src/reranker.rs).  They replace the ORT `session.run` of those modules; semantics follow oracle/bert_ref.py (pinned
// to transformers' BertForMaskedLM / BertForSequenceClassification).  The GEMMs are the ping-pong kernel of
// gemm_kernels.hip with a bias (+ erf-GELU) epilogue; this file holds what is BERT-specific: the three-table
// embedding + LayerNorm, residual + LayerNorm, multi-head attention for head dims 32 / 64 and the SPLADE activation
// (the pooling itself is the decoder GEMM's epilogue, launch_gemm_rowmax).
// Tokens are packed (no padding reaches a kernel); bf16 operands, f32 accumulation and statistics.
#include "bert_kernels.h"
#include "launch_util.h"
#include <stdlib.h>
#include <type_traits>

namespace cqs {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

namespace {

// cross-lane-group reductions of the attention's softmax by v_permlane{16,32}_swap (VALU) instead of ds_bpermute
// (LDS queue): lanes l, l^16, l^32, l^48 hold the same query's other keys
__device__ __forceinline__ float xg_max(float v) {
    typedef unsigned pu2 __attribute__((ext_vector_type(2)));
    const pu2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const pu2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xg_sum(float v) {
    typedef unsigned pu2 __attribute__((ext_vector_type(2)));
    const pu2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const pu2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int kMaxChunks = 8;   // hidden <= 1024

// A token row lives in one wave as chunks of VEC consecutive elements: chunk i of lane l = elements
// (64 i + l) * VEC ..+VEC-1 (VEC = 4: 8-byte accesses, hidden % 256 == 0; VEC = 2 otherwise, hidden % 128 == 0).
// LayerNorm: mean and biased variance in f32 (two passes over the registers), y = (x - mean) * rsqrt(var + eps) *
// gamma + beta (torch.nn.functional.layer_norm).
template <int VEC>
__device__ __forceinline__ void ln_row_store(float (&v)[kMaxChunks][VEC], int nc, uint32_t H, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float eps, bf16_t* __restrict__ out, int lane) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i)
        if (i < nc) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += v[i][e];
        }
    const float mean = wave_sum64(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i)
        if (i < nc) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float a = v[i][e] - mean; q += a * a; }
        }
    const float inv = rsqrtf(wave_sum64(q) / (float)H + eps);
    typedef __bf16 bfv __attribute__((ext_vector_type(VEC)));
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i)
        if (i < nc) {
            const uint32_t c = (uint32_t)((64 * i + lane) * VEC);
            bfv o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = (bf16_t)((v[i][e] - mean) * inv * gamma[c + e] + beta[c + e]);
            *(bfv*)(out + c) = o;
        }
}

// one wave per token: x = word[tok] + position[pos] + token_type[tt]; out = LN(x)
template <int VEC>
__global__ __launch_bounds__(256) void bert_embed_ln_kernel(const int32_t* __restrict__ tok, const int32_t* __restrict__ pos,
                                                            const int32_t* __restrict__ tt, const bf16_t* __restrict__ word,
                                                            const bf16_t* __restrict__ posw, const bf16_t* __restrict__ typew,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, bf16_t* __restrict__ out, uint32_t M, uint32_t H) {
    typedef __bf16 bfv __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const uint32_t m = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nc = (int)(H / (64u * VEC));
    const bf16_t* a = word + (size_t)tok[m] * H;
    const bf16_t* b = posw + (size_t)pos[m] * H;
    const bf16_t* c = typew + (size_t)tt[m] * H;
    float v[kMaxChunks][VEC];
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i)
        if (i < nc) {
            const uint32_t col = (uint32_t)((64 * i + lane) * VEC);
            const bfv x = *(const bfv*)(a + col), y = *(const bfv*)(b + col), z = *(const bfv*)(c + col);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[i][e] = (float)x[e] + (float)y[e] + (float)z[e];
        }
    ln_row_store<VEC>(v, nc, H, gamma, beta, eps, out + (size_t)m * H, lane);
}

// one wave per token: out = LN(a + r) (r == NULL: LN(a)); out may alias a
template <int VEC>
__global__ __launch_bounds__(256) void bert_add_ln_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ r,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, bf16_t* __restrict__ out, uint32_t M, uint32_t H) {
    typedef __bf16 bfv __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const uint32_t m = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nc = (int)(H / (64u * VEC));
    float v[kMaxChunks][VEC];
#pragma unroll
    for (int i = 0; i < kMaxChunks; ++i)
        if (i < nc) {
            const uint32_t col = (uint32_t)((64 * i + lane) * VEC);
            const bfv x = *(const bfv*)(a + (size_t)m * H + col);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[i][e] = (float)x[e];
            if (r) {
                const bfv y = *(const bfv*)(r + (size_t)m * H + col);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[i][e] += (float)y[e];
            }
        }
    ln_row_store<VEC>(v, nc, H, gamma, beta, eps, out + (size_t)m * H, lane);
}

// The same for the hidden sizes of the models at hand (768, 1024, 256; 384 with 4-byte lanes), TWO tokens per wave with every load of both
// in flight before the first reduction, chunk count at compile time, gamma / beta as 16-byte loads: the kernel is a pure
// HBM stream (75 MB at 64 x 256 tokens x 768) and ran at 65 % of the rate add_norm_kernel reaches.  Same per-lane order of
// additions and the same reductions as bert_add_ln_kernel<4>: bit-identical.
template <int VEC, int NC>
__global__ __launch_bounds__(256) void bert_add_ln2_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ r,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, bf16_t* __restrict__ out, uint32_t M) {
    typedef __bf16 bfv __attribute__((ext_vector_type(VEC)));
    typedef float fv __attribute__((ext_vector_type(VEC)));
    constexpr uint32_t H = 64u * VEC * NC;
    const int lane = threadIdx.x & 63;
    const uint32_t m0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2u;
    if (m0 >= M) return;
    const bool two = m0 + 1u < M;
    const uint32_t m1 = two ? m0 + 1u : m0;
    float v[2][NC][VEC];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const size_t row = (size_t)(t ? m1 : m0) * H;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const uint32_t col = (uint32_t)((64 * i + lane) * VEC);
            const bfv x = *(const bfv*)(a + row + col);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[t][i][e] = (float)x[e];
            if (r) {
                const bfv y = *(const bfv*)(r + row + col);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[t][i][e] += (float)y[e];
            }
        }
    }
    fv g[NC], b[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        g[i] = *(const fv*)(gamma + (64 * i + lane) * VEC);
        b[i] = *(const fv*)(beta + (64 * i + lane) * VEC);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) sm += v[t][i][e];
        const float mean = wave_sum64(sm) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v[t][i][e] - mean; q += d * d; }
        const float inv = rsqrtf(wave_sum64(q) / (float)H + eps);
        if (t == 1 && !two) break;
        bf16_t* o = out + (size_t)(t ? m1 : m0) * H;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            bfv w;
#pragma unroll
            for (int e = 0; e < VEC; ++e) w[e] = (bf16_t)((v[t][i][e] - mean) * inv * g[i][e] + b[i][e]);
            *(bfv*)(o + (64 * i + lane) * VEC) = w;
        }
    }
}

// ---- multi-head attention, head dim HD = 32 or 64, bidirectional over one packed sequence -----------------------
// One workgroup = 4 waves = 64 consecutive queries of ONE head of one sequence, 16 queries per wave.  Same product
// layout as attention_dma_kernel (embed_kernels.hip): S^T = K Q^T with keys on MFMA rows (softmax is lane-local), the
// S^T tile (t, kt) takes its key rows in the order 32 t + 8 (i >> 2) + 4 kt + (i & 3) so that a lane group's P values
// are the 8 consecutive keys 8 lg .. 8 lg + 7 = the PV product's k-indices, and the V^T fragments come from the
// ROW-major V tile by ds_read_b64_tr_b16 (lane 4 q + p of a 16-lane group addresses key 8 lg + 4 h + q, dims
// 16 dt + 4 p ..+3).  K / V tiles of 64 keys go through registers into LDS (rows padded by 16 B).
template <int HD>
__global__ __launch_bounds__(256) void bert_attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                             const int32_t* __restrict__ blk, const int32_t* __restrict__ seq_start,
                                                             const int32_t* __restrict__ seq_len, uint32_t heads, float scale) {
    constexpr int kRow = HD + 8;                       // LDS row stride (elements)
    constexpr int kCh = HD / 8;                        // 16-byte chunks per row
    constexpr int kU = 64 * kCh / 256;                 // chunks per thread per tile (1 or 2)
    constexpr int kST = HD / 32, kDT = HD / 16;
    __shared__ __attribute__((aligned(16))) bf16_t sK[64 * kRow];
    __shared__ __attribute__((aligned(16))) bf16_t sV[64 * kRow];
    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) tr4* lds_tr4;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const uint32_t b = (uint32_t)blk[2 * blockIdx.x], qb = (uint32_t)blk[2 * blockIdx.x + 1];
    const uint32_t head = blockIdx.y;
    const uint32_t s0 = (uint32_t)seq_start[b], L = (uint32_t)seq_len[b];
    const uint32_t H = heads * (uint32_t)HD, ld = 3u * H;
    const uint32_t q0 = qb * 64u + (uint32_t)wid * 16u;    // this wave's first query (the block exists: qb * 64 < L)
    const bool wave_live = q0 < L;
    const uint32_t qi = q0 + (uint32_t)l15;
    const uint32_t qrow = s0 + (qi < L ? qi : L - 1u);

    bf8 qf[kST];
#pragma unroll
    for (int s = 0; s < kST; ++s) qf[s] = *(const bf8*)(qkv + (size_t)qrow * ld + head * HD + 32 * s + 8 * lg);

    f4 o[kDT];
#pragma unroll
    for (int d = 0; d < kDT; ++d) o[d] = (f4)(0.f);
    float m_run = -INFINITY, l_run = 0.f;
    const float c = scale * 1.4426950408889634f;       // scores enter exp2 as s * c

    const uint32_t nkb = (L + 63u) / 64u;
    u4 rk[kU], rv[kU];
    auto stage_load = [&](uint32_t kb) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = (uint32_t)(u * 256 + tid), r = i / (uint32_t)kCh, ch = i % (uint32_t)kCh;
            uint32_t key = kb * 64u + r;
            key = key < L ? key : L - 1u;              // rows past the sequence: any real row (masked / P = 0)
            const bf16_t* p = qkv + (size_t)(s0 + key) * ld + head * HD + ch * 8u;
            rk[u] = *(const u4*)(p + H);
            rv[u] = *(const u4*)(p + 2u * H);
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = (uint32_t)(u * 256 + tid), r = i / (uint32_t)kCh, ch = i % (uint32_t)kCh;
            *(u4*)(sK + r * kRow + ch * 8u) = rk[u];
            *(u4*)(sV + r * kRow + ch * 8u) = rv[u];
        }
    };
    const uint32_t lds_v = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sV;
    const int tq = (lane >> 2) & 3, tp = lane & 3;

    stage_load(0);
    for (uint32_t kb = 0; kb < nkb; ++kb) {
        __syncthreads();                               // previous tile fully consumed
        stage_write();
        __syncthreads();
        if (kb + 1u < nkb) stage_load(kb + 1u);
        if (!wave_live) continue;                      // wave-uniform
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t k_first = kb * 64u + 32u * (uint32_t)t;
            if (k_first >= L) break;                   // wave-uniform
            f4 sc[2];
            sc[0] = (f4)(0.f);
            sc[1] = (f4)(0.f);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < kST; ++s) {
                    const int row = 32 * t + 8 * (l15 >> 2) + 4 * kt + (l15 & 3);
                    const bf8 kf = *(const bf8*)(sK + row * kRow + 32 * s + 8 * lg);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], sc[kt], 0, 0, 0);
                }
            // sc[kt][r] = q . k for key k_first + 8 lg + 4 kt + r
            float mloc = -INFINITY;
            if (k_first + 31u >= L) {                  // (wave-uniform) the half reaches past the sequence: mask key by key
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t key = k_first + (uint32_t)(8 * lg + 4 * kt + r);
                        sc[kt][r] = key < L ? sc[kt][r] : -INFINITY;
                    }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, sc[kt][r]);
            mloc = xg_max(mloc);
            const float m_new = fmaxf(m_run, mloc);    // finite: key k_first < L is attendable for every query
            const float a = __builtin_amdgcn_exp2f((m_run - m_new) * c);     // exp2(-inf) = 0 on the first half
            l_run *= a;
#pragma unroll
            for (int d = 0; d < kDT; ++d) o[d] *= a;
            m_run = m_new;
            bf8 pf;
            float lsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][r], c, -m_new * c));
                    lsum += pv;
                    pf[kt * 4 + r] = (bf16_t)pv;
                }
            l_run += lsum;                             // per-lane partial; reduced over the lane groups at the end
#pragma unroll
            for (int dt = 0; dt < kDT; ++dt) {
                const uint32_t ad = lds_v + (uint32_t)(((32 * t + 8 * lg + tq) * kRow + 16 * dt + 4 * tp) * 2);
                const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4)(uintptr_t)ad);
                const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4)(uintptr_t)(ad + (uint32_t)(4 * kRow * 2)));
                const bf8 vf = __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
            }
        }
    }
    l_run = xg_sum(l_run);
    if (wave_live && qi < L) {
        const float invl = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* op = out + (size_t)(s0 + qi) * H + head * HD;
#pragma unroll
        for (int d = 0; d < kDT; ++d) {
            bf4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(o[d][e] * invl);
            *(bf4*)(op + 16 * d + 4 * lg) = w;          // O^T[dim 16 d + 4 lg + r][query l15]
        }
    }
}

// ---- multi-head attention, second generation: a sequence's keys RESIDENT in LDS ----------------------------------
// One workgroup = NW waves = one (sequence, head) pair - or one of `qsplit` parts of its queries.  The head's K and V
// rows (HD * 2 bytes each, exactly as they lie in the qkv buffer) are fetched ONCE by LDS-DMA into images that hold the
// whole sequence (NG groups of 128 keys), group by group; the workgroup then walks its queries in passes of 16 NW (one
// 16-query tile per wave) over the resident keys.  The first pass starts on group 0 while groups 1.. are still in
// flight (counted vmcnt + one barrier per group, first pass only); after that there is no barrier and no wait left.
// Against bert_attention_kernel (64 queries / workgroup, 64-key tiles through registers, two barriers and one rescale of
// O per 32 keys, K / V re-read per 64 queries): the running maximum moves once per 128 KEYS (32 scores per lane: exact
// maximum first, then one exp per score), the row sums ride on the matrix cores (one more PV tile whose V^T rows are
// all ones: sum_k P[k][q], from the same bf16 P the numerator uses), key masking is a branch only the sequence's last
// group takes, no ds_write is issued, K / V cross L2 -> LDS once per (sequence, head).  The loop is VALU-bound by design:
// per 128 keys and wave 33 v_exp (quarter rate) + 32 fma + 16 max3 + 16 cvt_pk against 36..40 MFMAs.
// Same product layout as above (S^T = K Q^T, keys on MFMA rows in the order 32 t + 8 (i >> 2) + 4 kt + (i & 3); V^T
// fragments by ds_read_b64_tr_b16 from the row-major V image).  The images are unpadded; the 16-byte chunk c of key
// row r lives at position c ^ f(r) - applied to the DMA's SOURCE address - with
//   HD = 64 (128-byte rows):  f(r) = r3 | r1 << 1 | r3 << 2        HD = 32 (64-byte rows):  f(r) = r3 | (r3 ^ r4) << 1
// (r1, r3, r4: bits of r), so that the 16 rows of a ds_read_b128 lane group and the 8 rows x 32 B of a transposed
// read's 32-lane half fall into distinct bank slots (MI355X_MICROARCH.md, LDS lane groups).
template <int HD>
__device__ __forceinline__ uint32_t res_swz(uint32_t r) {
    if (HD == 64) return ((r >> 3) & 1u) * 5u ^ (((r >> 1) & 1u) << 1);
    return ((r >> 3) & 1u) | ((((r >> 3) ^ (r >> 4)) & 1u) << 1);
}

template <int HD, int NW, int NG>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void bert_attention_res_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, const int32_t* __restrict__ seq_start,
    const int32_t* __restrict__ seq_len, uint32_t heads, uint32_t qsplit, float scale) {
    constexpr int kST = HD / 32, kDT = HD / 16;
    constexpr int kCh = HD / 8;                                   // 16-byte chunks per row
    constexpr int kRowsPerDma = 1024 / (HD * 2);                  // key rows one DMA instruction fills (8 or 16)
    constexpr int kDmaPerGroup = 128 / kRowsPerDma;               // instructions per 128-key group of ONE image
    constexpr int kPerWave = kDmaPerGroup / NW;                   // ... per wave (K; as many again for V)
    static_assert(kDmaPerGroup % NW == 0 && kPerWave >= 1, "a group's fetch splits evenly over the waves");
    constexpr int kOut = 2 * kPerWave;                            // DMA instructions in flight per wave and group
    constexpr uint32_t kImg = (uint32_t)NG * 128u * HD * 2u;      // bytes of one image (K or V)
    extern __shared__ __attribute__((aligned(1024))) bf16_t res_smem[];     // K image | V image
    typedef short tr4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) tr4* lds_tr4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    // one contiguous run of the (sequence, head, part) list per XCD: the parts of one pair share an L2
    const uint32_t nwg = gridDim.x, xcd = blockIdx.x % 8u, q8 = nwg / 8u, r8 = nwg % 8u;
    const uint32_t wg = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + blockIdx.x / 8u;
    const uint32_t part = wg % qsplit, pair = wg / qsplit;
    const uint32_t b = pair / heads, head = pair % heads;
    const uint32_t s0 = (uint32_t)seq_start[b], L = (uint32_t)seq_len[b];
    const uint32_t npass = (L + 16u * NW - 1u) / (16u * NW);
    if (part >= npass) return;                                    // (uniform, before any barrier; covers L == 0)
    const uint32_t ng = (L + 127u) / 128u;                        // <= NG: the host checked the longest sequence
    const uint32_t H = heads * (uint32_t)HD, ld = 3u * H;

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)res_smem;
    const char* const gK = (const char*)(qkv + (size_t)s0 * ld + H + head * HD);
    const uint32_t vrel = H * 2u;                                 // byte distance from a token's k head to its v head
    auto dma = [&](uint32_t voff, uint32_t lds_byte) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds_byte), "v"(voff), "s"(gK) : "memory");
    };
    // group g: instruction i (1 KiB of each image) by wave i % NW; rows past the sequence read its last key (finite: V
    // times P = 0; their scores are masked)
    auto stage = [&](uint32_t g) {
#pragma unroll
        for (int j = 0; j < kPerWave; ++j) {
            const uint32_t i = g * (uint32_t)kDmaPerGroup + (uint32_t)wid + (uint32_t)(NW * j);
            const uint32_t r = (uint32_t)kRowsPerDma * i + (uint32_t)lane / (uint32_t)kCh;
            const uint32_t key = r < L ? r : L - 1u;
            const uint32_t c = ((uint32_t)lane % (uint32_t)kCh) ^ res_swz<HD>(r);
            const uint32_t voff = key * ld * 2u + c * 16u;
#if defined(CQS_BATT_NO_DMA)
            if (voff == 0xFFFFFFFFu)
#endif
            {
                dma(voff, lds0 + i * 1024u);
                dma(voff + vrel, lds0 + kImg + i * 1024u);
            }
        }
    };
    // at most k groups of this wave's fetches still in flight
    auto wait_groups = [&](uint32_t k) {
        if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kOut) : "memory");
        else if (k == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kOut) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * kOut) : "memory");
    };
    static_assert(NG <= 4, "wait_groups covers four groups");

    // fragment addresses.  K: row(l15) of tile (t, kt) = 32 t + 4 kt + 8 (l15 >> 2) + (l15 & 3), chunk 4 s + lg
    const uint32_t krow = (uint32_t)(8 * (l15 >> 2) + (l15 & 3));
    const bf16_t* kp[kST];
#pragma unroll
    for (int s = 0; s < kST; ++s) kp[s] = res_smem + krow * (uint32_t)HD + (((uint32_t)(4 * s + lg)) ^ res_swz<HD>(krow)) * 8u;
    // V (transposed reads): lane 16 lg + 4 q + p addresses key row 8 lg + q (+ 4 h + 32 t), dims 16 dt + 4 p ..+3
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    const uint32_t vrow = (uint32_t)(8 * lg + tq);
    uint32_t vb[kDT];
#pragma unroll
    for (int dt = 0; dt < kDT; ++dt)
        vb[dt] = lds0 + kImg + vrow * (uint32_t)(HD * 2) + ((((uint32_t)(2 * dt + (tp >> 1))) ^ res_swz<HD>(vrow)) * 16u) +
                 (uint32_t)(tp & 1) * 8u;
    bf8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    const float c = scale * 1.4426950408889634f;                  // scores enter exp2 as s * c

    uint32_t q0 = part * (16u * NW) + (uint32_t)wid * 16u;        // this wave's first query of the pass
    // first pass: group 0, then the Q fragments, then the other groups - in that order in the vmcnt queue, so that group 0
    // and Q can be waited for with groups 1.. still in flight.  (The Q loads are asm: tracked by hipcc they would get a
    // vmcnt(0) at their first use, which also drains every DMA; nothing touches qf between the loads and the wait.)
    stage(0u);
    bf8 qf[kST];
    {
        const uint32_t qi = q0 + (uint32_t)l15;
        const bf16_t* qp = qkv + (size_t)(s0 + (qi < L ? qi : L - 1u)) * ld + head * HD + 8 * lg;
#pragma unroll
        for (int s = 0; s < kST; ++s)
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(qf[s]) : "v"(qp), "n"(64 * s) : "memory");
    }
    for (uint32_t g = 1; g < ng; ++g) stage(g);
    wait_groups(ng - 1u);                                         // this wave's share of group 0, and its Q, have landed
#pragma unroll
    for (int s = 0; s < kST; ++s) asm volatile("" : "+v"(qf[s]));
    __syncthreads();                                              // everyone's share of group 0 has

    for (uint32_t pi = 0;; ++pi) {
        const bool wave_live = q0 < L;                            // waves past the sequence only help staging
        const uint32_t qi = q0 + (uint32_t)l15;
        f4 o[kDT], ls = (f4)(0.f);
#pragma unroll
        for (int d = 0; d < kDT; ++d) o[d] = (f4)(0.f);
        float m_run = -INFINITY;

        // 128 keys at LDS rows 128 g ..
        auto group = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            const uint32_t kfirst = 128u * g;
            f4 sc[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    sc[t][kt] = (f4)(0.f);
#pragma unroll
                    for (int s = 0; s < kST; ++s) {
#if defined(CQS_BATT_NO_KREAD)
                        const bf8 kf = qf[(s + t) % kST];
#else
                        const bf8 kf = *(const bf8*)(kp[s] + (128 * g + 32 * t + 4 * kt) * HD);
#endif
#if defined(CQS_BATT_NO_S)
                        sc[t][kt][s] += (float)kf[0];
#else
                        sc[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], sc[t][kt], 0, 0, 0);
#endif
                    }
                }
            // sc[t][kt][r] = q . k for key kfirst + 32 t + 8 lg + 4 kt + r
            if (kfirst + 127u >= L) {                             // (wave-uniform) the sequence's last, partial group
                int lim = (int)L - (int)kfirst - 8 * lg;
                asm volatile("" : "+v"(lim));                     // keep it a branch (full groups pay nothing), and keep the
                                                                  // 32 compare masks out of the pass loop's SGPRs
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[t][kt][r] = (32 * t + 4 * kt + r) < lim ? sc[t][kt][r] : -INFINITY;
            }
            float mloc = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, sc[t][kt][r]);
            mloc = xg_max(mloc);
            const float m_new = fmaxf(m_run, mloc);               // finite: key kfirst < L is attendable for every query
            const float a = __builtin_amdgcn_exp2f((m_run - m_new) * c);      // exp2(-inf) = 0 on the first group
            ls *= a;
#pragma unroll
            for (int d = 0; d < kDT; ++d) o[d] *= a;
            m_run = m_new;
            const float mc = -m_new * c;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf8 pf;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#if defined(CQS_BATT_NO_EXP)
                        pf[kt * 4 + r] = (bf16_t)__builtin_fmaf(sc[t][kt][r], c, mc);
#else
                        pf[kt * 4 + r] = (bf16_t)__builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][kt][r], c, mc));
#endif
                ls = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, ls, 0, 0, 0);     // every row: sum_k P[k][query]
#pragma unroll
                for (int dt = 0; dt < kDT; ++dt) {
#if defined(CQS_BATT_NO_VREAD)
                    const bf8 vf = qf[dt % kST];
#else
                    const uint32_t ad = vb[dt] + (uint32_t)((128 * g + 32 * t) * HD * 2);
                    const tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4)(uintptr_t)ad);
                    const tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4)(uintptr_t)(ad + (uint32_t)(4 * HD * 2)));
                    const bf8 vf = __builtin_bit_cast(bf8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#endif
#if defined(CQS_BATT_NO_PV)
                    o[dt][0] += (float)vf[0] * (float)pf[dt];
#else
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
#endif
                }
            }
        };
        auto step = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            if constexpr (g < NG) {
                if ((uint32_t)g < ng) {                           // (uniform)
                    if (g > 0 && pi == 0) {
                        wait_groups(ng - 1u - (uint32_t)g);       // this wave's share of group g has landed
                        __syncthreads();                          // everyone's has
                    }
                    if (wave_live) group(g_c);
                }
            }
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});

#if defined(CQS_BATT_NO_STORE)
        if (wave_live && qi < L && o[0][0] == 1234.5f) {
#else
        if (wave_live && qi < L) {
#endif
            const float invl = ls[0] > 0.f ? 1.0f / ls[0] : 0.f;
            bf16_t* op = out + (size_t)(s0 + qi) * H + head * HD;
            // O^T[dim 16 d + 4 lg + e][query l15]: a lane holds 8-byte pieces 32 bytes apart.  One v_permlane16_swap per
            // register of a tile pair (d, d + 1) turns them into 16 contiguous bytes per lane (lane groups 0 / 2 keep
            // tile d's dims 8 lg' .. + 7, groups 1 / 3 tile d + 1's), so a store instruction writes 64 contiguous bytes
            // of each of its 16 rows instead of 32
            typedef unsigned pu2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int d = 0; d < kDT; d += 2) {
                bf4 w0, w1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { w0[e] = (bf16_t)(o[d][e] * invl); w1[e] = (bf16_t)(o[d + 1][e] * invl); }
                const pu2 a = __builtin_bit_cast(pu2, w0), b2 = __builtin_bit_cast(pu2, w1);
                const pu2 x = __builtin_amdgcn_permlane16_swap(a[0], b2[0], false, false);
                const pu2 y = __builtin_amdgcn_permlane16_swap(a[1], b2[1], false, false);
                // even lane groups: (own tile d, the next group's tile d); odd: (the previous group's tile d + 1, own)
                u4 v;
                v[0] = x[0]; v[1] = y[0]; v[2] = x[1]; v[3] = y[1];
                const int col = 16 * d + ((lg & 1) ? 16 : 0) + 8 * (lg >> 1);
                *(u4*)(op + col) = v;
            }
        }
        q0 += qsplit * (16u * NW);
        if (q0 - (uint32_t)wid * 16u >= L) break;                 // (uniform: the pass's first query) no pass left
        // the next pass's Q fragments, waited for right here (by now nothing else is in flight; a wait that hipcc placed
        // at their first use would sit in front of group 0 and, in the first pass, drain every DMA): the other waves
        // of the SIMD cover the round trip
        const uint32_t qx = q0 + (uint32_t)l15;
        const bf16_t* qp = qkv + (size_t)(s0 + (qx < L ? qx : L - 1u)) * ld + head * HD + 8 * lg;
#pragma unroll
        for (int s = 0; s < kST; ++s) qf[s] = *(const bf8*)(qp + 32 * s);
#pragma unroll
        for (int s = 0; s < kST; ++s) asm volatile("" : "+v"(qf[s]));
    }
}

// SPLADE activation (src/splade/mod.rs:1049-1053) over the pooled maxima the decoder GEMM left behind
// (launch_gemm_rowmax: max(0, max_s logits), NaN never taken): x <- ln(1 + x), in place.
__global__ __launch_bounds__(256) void splade_activate_kernel(float* __restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) x[i] = logf(1.0f + x[i]);
}

// Threshold filter of src/splade/mod.rs:1049-1062 on the device: row b of the [B, V] activations -> its entries
// > threshold as (id, weight), ascending id, at most `cap` of them (count[b] reports how many there ARE: a count above
// cap tells the caller to take the dense row instead).  One workgroup of 16 waves per row; a thread owns 32 CONSECUTIVE
// columns of a 32 768-column step (a BERT vocabulary is one step: every load of the row is in flight at once, one
// block-wide exclusive scan of the per-thread hit counts orders the output; the first version - 256 threads, 4 096
// columns and two barriers per step - took 26 us for the one row of a search-time query).  NaN > t is false (dropped),
// +Inf passes, as in the reference.
__global__ __launch_bounds__(1024) void splade_sparsify_kernel(const float* __restrict__ dense, uint32_t V, float threshold,
                                                               uint32_t cap, uint32_t* __restrict__ out_ids,
                                                               float* __restrict__ out_w, uint32_t* __restrict__ out_count) {
    constexpr uint32_t PER = 32;
    __shared__ uint32_t wave_tot[16];
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const float* row = dense + (size_t)b * V;
    uint32_t base = 0;                                   // hits in the steps before this one (the same in every thread)
    for (uint32_t v0 = 0; v0 < V; v0 += 1024u * PER) {
        const uint32_t v = v0 + tid * PER;
        float x[PER];
        uint32_t hit = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) x[j] = v + j < V ? row[v + j] : 0.f;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) hit |= (uint32_t)(v + j < V && x[j] > threshold) << j;
        const uint32_t n = (uint32_t)__builtin_popcount(hit);
        // exclusive scan of n over the workgroup: inside the wave by shuffles, across the 16 waves through LDS
        uint32_t inc = n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += t;
        }
        if (lane == 63u) wave_tot[wid] = inc;
        __syncthreads();
        uint32_t before = base, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16u; ++w) {
            const uint32_t t = wave_tot[w];
            before += w < wid ? t : 0u;
            total += t;
        }
        uint32_t pos = before + inc - n;
        if (hit) {
#pragma unroll
            for (uint32_t j = 0; j < PER; ++j)
                if ((hit >> j) & 1u) {
                    if (pos < cap) { out_ids[(size_t)b * cap + pos] = v + j; out_w[(size_t)b * cap + pos] = x[j]; }
                    ++pos;
                }
        }
        base += total;
        __syncthreads();                                 // wave_tot is rewritten by the next step
    }
    if (tid == 0) out_count[b] = base;
}

// Pooling of the BERT-family EMBEDDERS (e5-base, v9-200k, bge-large presets: src/embedder/models.rs:346-405), over the
// packed final hidden states: mode 0 = `mean_pool` (src/embedder/pooling.rs:87-121: sum over the sequence's tokens /
// their count, an empty sequence gives zeros), mode 1 = `cls_pool` (:123-128: the first token).  One workgroup per
// sequence, a thread per 4 columns; f32 accumulation in token order.
__global__ __launch_bounds__(256) void bert_pool_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ seq_start,
                                                        const int32_t* __restrict__ seq_len, float* __restrict__ out,
                                                        uint32_t H, int mode) {
    const uint32_t b = blockIdx.x, c = threadIdx.x * 4u;
    if (c >= H) return;
    const uint32_t s0 = (uint32_t)seq_start[b], L = (uint32_t)seq_len[b];
    f4 acc = (f4)(0.f);
    const uint32_t n = mode == 1 ? (L ? 1u : 0u) : L;
    for (uint32_t t = 0; t < n; ++t) {
        const bf4 v = *(const bf4*)(x + (size_t)(s0 + t) * H + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (float)v[e];
    }
    if (mode == 0 && L) {
        const float cnt = (float)L;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] / cnt;
    }
    *(f4*)(out + (size_t)b * H + c) = acc;
}

}  // namespace

hipError_t launch_bert_pool(const bf16_t* x, const int32_t* seq_start, const int32_t* seq_len, float* out, uint32_t B,
                            uint32_t H, int mode, hipStream_t st) {
    if (B == 0) return hipSuccess;
    if (H % 4u || H > 1024u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bert_pool_kernel, dim3(B), dim3(256), 0, st, x, seq_start, seq_len, out, H, mode);
    return hipGetLastError();
}

hipError_t launch_splade_sparsify(const float* dense, uint32_t B, uint32_t V, float threshold, uint32_t cap, uint32_t* out_ids,
                                  float* out_w, uint32_t* out_count, hipStream_t st) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(splade_sparsify_kernel, dim3(B), dim3(1024), 0, st, dense, V, threshold, cap, out_ids, out_w, out_count);
    return hipGetLastError();
}

hipError_t launch_bert_embed_ln(const int32_t* tok, const int32_t* pos, const int32_t* tt, const bf16_t* word,
                                const bf16_t* posw, const bf16_t* typew, const float* gamma, const float* beta, float eps,
                                bf16_t* out, uint32_t M, uint32_t H, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (H % 128u || H > 1024u) return hipErrorInvalidValue;
    if (H % 256u == 0)
        hipLaunchKernelGGL(bert_embed_ln_kernel<4>, dim3((M + 3u) / 4u), dim3(256), 0, st, tok, pos, tt, word, posw, typew, gamma,
                           beta, eps, out, M, H);
    else
        hipLaunchKernelGGL(bert_embed_ln_kernel<2>, dim3((M + 3u) / 4u), dim3(256), 0, st, tok, pos, tt, word, posw, typew, gamma,
                           beta, eps, out, M, H);
    return hipGetLastError();
}

hipError_t launch_bert_add_ln(const bf16_t* a, const bf16_t* r, const float* gamma, const float* beta, float eps,
                              bf16_t* out, uint32_t M, uint32_t H, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (H % 128u || H > 1024u) return hipErrorInvalidValue;
    const uint32_t g2 = (M + 7u) / 8u;                     // 4 waves x 2 tokens per workgroup
    if (H == 768u) hipLaunchKernelGGL((bert_add_ln2_kernel<4, 3>), dim3(g2), dim3(256), 0, st, a, r, gamma, beta, eps, out, M);
    else if (H == 1024u) hipLaunchKernelGGL((bert_add_ln2_kernel<4, 4>), dim3(g2), dim3(256), 0, st, a, r, gamma, beta, eps, out, M);
    else if (H == 256u) hipLaunchKernelGGL((bert_add_ln2_kernel<4, 1>), dim3(g2), dim3(256), 0, st, a, r, gamma, beta, eps, out, M);
    else if (H == 384u) hipLaunchKernelGGL((bert_add_ln2_kernel<2, 3>), dim3(g2), dim3(256), 0, st, a, r, gamma, beta, eps, out, M);
    else if (H % 256u == 0) hipLaunchKernelGGL(bert_add_ln_kernel<4>, dim3((M + 3u) / 4u), dim3(256), 0, st, a, r, gamma, beta, eps, out, M, H);
    else hipLaunchKernelGGL(bert_add_ln_kernel<2>, dim3((M + 3u) / 4u), dim3(256), 0, st, a, r, gamma, beta, eps, out, M, H);
    return hipGetLastError();
}

namespace {
template <int HD, int NW, int NG>
hipError_t launch_res(const bf16_t* qkv, bf16_t* out, const int32_t* seq_start, const int32_t* seq_len, uint32_t B,
                      uint32_t max_len, uint32_t heads, float scale, hipStream_t st) {
    static DynLdsOnce once;
    auto kern = bert_attention_res_kernel<HD, NW, NG>;
    constexpr size_t lds = (size_t)2 * NG * 128 * HD * 2;
    hipError_t e = once.ensure((const void*)kern, lds);
    if (e != hipSuccess) return e;
    // how many workgroups share one (sequence, head): each loads the whole K / V, so as few as fill the chip evenly.
    // Model: workgroups on the busiest CU x (passes per workgroup + ~half a pass of exposed fetch)
    const uint32_t pairs = B * heads, slots = 256u;             // (a CU's second workgroup shares its VALU: count CUs)
    const uint32_t npass = (max_len + 16u * NW - 1u) / (16u * NW);
    uint32_t qsplit = 1;
    if (const char* v = getenv("CQS_HIP_BERT_ATTN_QSPLIT")) qsplit = (uint32_t)atoi(v);
    else {
        float best = 0.f;
        for (uint32_t q = 1; q <= npass; q *= 2u) {
            const uint32_t rounds = (pairs * q + slots - 1u) / slots;
            const float cost = (float)rounds * ((float)((npass + q - 1u) / q) + 0.5f);
            if (q == 1u || cost < best) { best = cost; qsplit = q; }
        }
    }
    if (qsplit < 1u) qsplit = 1u;
    if (qsplit > npass) qsplit = npass;
    hipLaunchKernelGGL(kern, dim3(pairs * qsplit), dim3(64 * NW), lds, st, qkv, out, seq_start, seq_len, heads, qsplit, scale);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_bert_attention(const bf16_t* qkv, bf16_t* out, const int32_t* blk, uint32_t nblk, const int32_t* seq_start,
                                 const int32_t* seq_len, uint32_t B, uint32_t max_len, uint32_t heads, uint32_t head_dim,
                                 hipStream_t st) {
    if (nblk == 0 || B == 0) return hipSuccess;
    const float scale = 1.0f / sqrtf((float)head_dim);
    const char* v = getenv("CQS_HIP_BERT_ATTN_RESIDENT");         // read per batch: a test flips it inside one process
    const bool resident = !(v && v[0] == '0');
    if (resident && max_len <= 512u && (head_dim == 64u || head_dim == 32u)) {
        if (head_dim == 64u) {
            if (max_len <= 256u) return launch_res<64, 8, 2>(qkv, out, seq_start, seq_len, B, max_len, heads, scale, st);
            return launch_res<64, 16, 4>(qkv, out, seq_start, seq_len, B, max_len, heads, scale, st);
        }
        if (max_len <= 256u) return launch_res<32, 8, 2>(qkv, out, seq_start, seq_len, B, max_len, heads, scale, st);
        return launch_res<32, 8, 4>(qkv, out, seq_start, seq_len, B, max_len, heads, scale, st);
    }
    if (head_dim == 64u)
        hipLaunchKernelGGL(bert_attention_kernel<64>, dim3(nblk, heads), dim3(256), 0, st, qkv, out, blk, seq_start, seq_len, heads, scale);
    else if (head_dim == 32u)
        hipLaunchKernelGGL(bert_attention_kernel<32>, dim3(nblk, heads), dim3(256), 0, st, qkv, out, blk, seq_start, seq_len, heads, scale);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_splade_activate(float* x, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(splade_activate_kernel, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, st, x, n);
    return hipGetLastError();
}

}  // namespace cqs
