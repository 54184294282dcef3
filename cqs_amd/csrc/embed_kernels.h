// embed_kernels.h — launch interface of the gfx950 EmbeddingGemma forward kernels.
// Internal to libcqs_hip.so (public boundary: include/cqs_hip.h, embed section).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include "lane_swap.h"

namespace cqs {

typedef __bf16 bf16_t;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE; a process may run engines on several GPUs.  One bit per
// device ordinal in a per-call-site mask: the attribute is set the first time a kernel is launched on each device.
inline hipError_t set_max_dynamic_lds(const void* kernel, size_t bytes, std::atomic<uint64_t>& done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (dev < 64 && (done_mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev < 64) done_mask.fetch_or(bit, std::memory_order_release);
    return e;
}

// Sum over the 32 lanes of this lane's half-wave, the same bits in all of them, VALU only: four DPP steps inside each
// 16-lane row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then one v_permlane16_swap between the two rows of the
// half.  (As __shfl_xor these were 5 ds_bpermute round trips through the LDS queue per sum.)  ONE definition: the row
// norms of add_norm_kernel, of the pair-split fused kernel and of the QKV epilogue must add in the same order.
#if defined(__HIPCC__)
template <int CTRL>
__device__ __forceinline__ float half_sum_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_wave_sum32(float v) {
    v += half_sum_dpp<0xB1>(v);
    v += half_sum_dpp<0x4E>(v);
    v += half_sum_dpp<0x141>(v);
    v += half_sum_dpp<0x140>(v);
    const lane_u2 a = swap16_self(__builtin_bit_cast(unsigned, v));
    const unsigned even = a[0], odd = a[1];                   // locals: see lane_swap.h on bit_cast of a vector element
    return __builtin_bit_cast(float, even) + __builtin_bit_cast(float, odd);
}
#endif

// Model geometry (Gemma3 text encoder + sentence-transformers head); see
// oracle/gemma3_ref.py for the semantics each field drives.
struct EmbedGeom {
    uint32_t vocab, hidden, layers, heads, kv_heads, head_dim, inter, dense_hidden;
    uint32_t window;          // bidirectional sliding window: |q - k| < window (config window // 2 + 1)
    uint32_t sliding_pattern; // layer i is full attention iff (i + 1) % pattern == 0
    uint32_t max_seq;
    float rms_eps, theta_global, theta_local, q_scale;  // q_scale = query_pre_attn_scalar^-0.5
};

// x[m] = emb[tok[m]] * scale (f32 residual stream); xn[m] = bf16(rmsnorm(x[m]) * (1 + w_in))
hipError_t launch_embed_norm(const int32_t* tok, const bf16_t* emb, float scale, const float* w_in, float eps,
                             float* x, bf16_t* xn, uint32_t M, uint32_t H, hipStream_t st);

// x[m] += rmsnorm(y[m]) * (1 + w_post)  (y = the branch's GEMM output, bf16); then the next pre-norm of the new x:
//   final == 0: xn[m]  = bf16(rmsnorm(x[m]) * (1 + w_next))
//   final == 1: out[m] = f32 (rmsnorm(x[m]) * (1 + w_next))      (the model's final norm)
hipError_t launch_add_norm(float* x, const bf16_t* y, const float* w_post, const float* w_next, float eps,
                           bf16_t* xn, float* out, int final, uint32_t M, uint32_t H, hipStream_t st);

// launch_gemm_bf16(A, W, y, GEMM_OUT_BF16) + launch_add_norm(x, y, ...) in ONE launch (gemm_rowfuse.hip): a workgroup owns
// 64 whole rows x all H = 768 columns, so y never leaves the CU.  Same bits as the two-launch chain.  H must be 768.
bool gemm_addnorm_supported(uint32_t M, uint32_t H, uint32_t K);     // shape only (the engine decides by batch size: embedder.hip)
hipError_t launch_gemm_addnorm(const bf16_t* A, const bf16_t* W, float* x, const float* w_post, const float* w_next,
                               float eps, bf16_t* xn, float* out, int final, uint32_t M, uint32_t H, uint32_t K,
                               hipStream_t st);

// The same fused projection, second generation: a PAIR of workgroups owns 128 rows, each one 384-column half; row sums
// are exchanged between the partners inside the launch (gemm_rowfuse.hip).  xch: gemm_addnorm_pair_scratch_bytes(M) of
// device memory private to the stream; tag: differs from every earlier launch on that buffer, never 0; *err is set to 1
// if an exchange timed out (the results of that launch are then garbage; the launch itself always ends).
// Wp: the projection's weights [H][K] re-ordered once by launch_pack_rowfuse_w into the 2 KB blocks the kernel streams.
size_t gemm_addnorm_pair_scratch_bytes(uint32_t M);
hipError_t launch_pack_rowfuse_w(const bf16_t* src, bf16_t* dst, uint32_t N, uint32_t K, hipStream_t st);
hipError_t launch_gemm_addnorm_pair(const bf16_t* A, const bf16_t* Wp, float* x, const float* w_post, const float* w_next,
                                    float eps, bf16_t* xn, float* out, int final, uint32_t M, uint32_t H, uint32_t K,
                                    void* xch, uint32_t tag, unsigned* err, hipStream_t st);

// C[M,N] = A[M,K] (bf16, row-major) x W[N,K]^T (bf16, row-major), f32 accumulate on the matrix cores.
//   GEMM_OUT_BF16 : C bf16 [M, ldc]
//   GEMM_OUT_F32  : C f32  [M, ldc]
//   GEMM_OUT_GEGLU: W rows are interleaved per 64: 32 gate rows then the 32 up rows of the same
//                   channels; C bf16 [M, ldc] gets N/2 columns = gelu_tanh(gate) * up
// Requires N % 128 == 0, K % 64 == 0.
enum GemmOut { GEMM_OUT_BF16 = 0, GEMM_OUT_F32 = 1, GEMM_OUT_GEGLU = 2, GEMM_OUT_BF16_GELU = 3 /* bf16(gelu_erf(x + bias)): launch_gemm_p8 / _bias only */,
               GEMM_OUT_ROWMAX = 4 /* launch_gemm_rowmax only */, GEMM_OUT_QKV = 5 /* launch_gemm_qkv_rope only */,
               GEMM_OUT_GEGLU4 = 6 /* the 256-row kernel's GeGLU with gate / up rows interleaved per 4 (launch_permute_geglu_rows): the
                                     pairing happens in registers (one v_permlane16_swap per two accumulators), bf16 stage */ };

// What the QKV projection's fused epilogue needs (GEMM_OUT_QKV): per-head RMSNorm (1 + w), RoPE and the q scale applied
// to the tile before it is stored - the work of kv_prep_kernel (k heads) and of the attention kernels' Q prologue.
struct QkvEpilogue {
    const int32_t* pos;       // [M] position of each packed token in its sequence
    const float* wq;          // [256] q-head norm weight
    const float* wk;          // [256] k-head norm weight
    const float* cos_sin;     // [max_seq][128][2] of the layer type
    float eps, q_scale;
    uint32_t heads, kv_heads; // heads == 3 * kv_heads
};
// qkv[M, (heads + 2 kv) * 256] = norm / rope / scale (A W^T) in ONE launch.  tn = 5: 256 x 320 tiles, each = one whole q or k
// head (256 columns: normalised, rotated, q scaled) + a 64-column slice of v (stored as is), W = the projection's rows in
// tile order (launch_permute_qkv_rows); tn = 4: 256 x 256 tiles = one head or 256 columns of v each, W in natural order.
// Replaces GEMM + kv_prep + the attention kernel's own Q norm (attention then runs with q_norm_w = NULL).
// gemm_qkv_rope_tile: 5 / 4 when the geometry fits AND the planner would run that tile over the whole projection anyway
// (full rounds at this M), else 0 - the three-launch chain on smaller kernels is then the faster one.
int gemm_qkv_rope_tile(uint32_t M, uint32_t hidden, uint32_t heads, uint32_t kv_heads, uint32_t head_dim);
hipError_t launch_permute_qkv_rows(const bf16_t* wqkv, bf16_t* wf, uint32_t heads, uint32_t kv_heads, uint32_t K, hipStream_t st);
hipError_t launch_gemm_qkv_rope(const bf16_t* A, const bf16_t* W, bf16_t* qkv, uint32_t M, uint32_t K, int tn, const QkvEpilogue& epi,
                                hipStream_t st);
hipError_t launch_gemm_bf16(const bf16_t* A, const bf16_t* W, void* C, uint32_t M, uint32_t N, uint32_t K,
                            uint32_t ldc, GemmOut out, hipStream_t st, const float* bias = nullptr /*[N] f32; not with GEGLU*/,
                            const bf16_t* W_geglu4 = nullptr /*GEGLU: the same rows interleaved per 4 (launch_permute_geglu_rows):
                                                               the parts the 256-row kernel takes then pair gate / up in registers*/);
// dst = the GeGLU projection's rows (interleaved per 32: 32 gate rows, the same channels' 32 up rows) re-interleaved per 4:
// every 16-row tile = [gate c..c+3 | up c..c+3 | gate c+4..c+7 | up c+4..c+7].  Any multiple of 64 rows holds the same
// channels in both orders, so a projection may be cut there between kernels that read different orders.
hipError_t launch_permute_geglu_rows(const bf16_t* src, bf16_t* dst, uint32_t N, uint32_t K, hipStream_t st);

// A few rows against a big matrix (the Dense head: M = sequences of the batch): one workgroup per 16 x 16 output tile,
// K split over its 4 waves.  GEMM_OUT_BF16 / GEMM_OUT_F32; N % 16 == 0, K % 32 == 0; M > 256 goes to launch_gemm_bf16.
hipError_t launch_gemm_skinny(const bf16_t* A, const bf16_t* W, void* C, uint32_t M, uint32_t N, uint32_t K,
                              uint32_t ldc, GemmOut out, hipStream_t st);

// The skinny kernel on rows `lda` elements apart, + bias (nullable) and optional tanh; any M (the BERT heads).
hipError_t launch_gemm_rows(const bf16_t* A, uint32_t lda, const bf16_t* W, const float* bias, int act_tanh, void* C,
                            uint32_t M, uint32_t N, uint32_t K, uint32_t ldc, GemmOut out, hipStream_t st,
                            const int32_t* row_index = nullptr /*device: output row m reads A row row_index[m]*/);

// Second-generation kernel (gemm_kernels.hip): 256 x (64 tn) x 64 tiles, 8 waves in two rows that alternate load and
// multiply intervals, counted-vmcnt LDS-DMA.  tn in {3, 4, 5}; N % (64 tn) == 0, K % 64 == 0.
// launch_gemm_bf16 picks between it and the 128 x 128 kernel by shape.
hipError_t launch_gemm_p8(const bf16_t* A, const bf16_t* W, void* C, uint32_t M, uint32_t N, uint32_t K,
                          uint32_t ldc, GemmOut out, int tn, hipStream_t st, const float* bias = nullptr /*[N], f32*/);
// C = act(A W^T + bias) for the BERT-family engines (bias may be NULL): picks the tile width among those that divide N
// (N % 192 == 0 at least), 256-row ping-pong kernel only.  out: GEMM_OUT_BF16, GEMM_OUT_BF16_GELU or GEMM_OUT_F32.
hipError_t launch_gemm_bias(const bf16_t* A, const bf16_t* W, const float* bias, void* C, uint32_t M, uint32_t N,
                            uint32_t K, uint32_t ldc, GemmOut out, hipStream_t st);

// Per-sequence column maxima of max(0, bf16(A W^T + bias)) without storing the product (SPLADE decoder + pooling):
// out_bits[row_seq[m]][v] = max(.., float bits), v < n_valid; out_bits [sequences, ldc] u32 zeroed by the caller.
hipError_t launch_gemm_rowmax(const bf16_t* A, const bf16_t* W, const float* bias, uint32_t* out_bits, uint32_t M, uint32_t N,
                              uint32_t K, uint32_t ldc, const int32_t* row_seq, uint32_t n_valid, hipStream_t st);

// Two column ranges of one GEMM (same A, M, K, ldc; different tile widths) in one launch.  hipErrorNotSupported when
// the pair of widths is not built: launch the parts one after the other instead.
hipError_t launch_gemm_p8_dual(const bf16_t* A, const bf16_t* Wa, void* Ca, uint32_t Na, int tn_a, const bf16_t* Wb,
                               void* Cb, uint32_t Nb, int tn_b, uint32_t M, uint32_t K, uint32_t ldc, GemmOut out,
                               hipStream_t st);

// In place on qkv [M, (heads + 2 kv) * 256] bf16: per-head RMSNorm * (1 + w), RoPE from the
// cos/sin table of the layer type, q additionally scaled by q_scale.  pos[m] = position in sequence.
// k_only != 0: the k heads only (launch_attention then does the same for its own Q fragments, q_norm_w != NULL).
hipError_t launch_qk_norm_rope(bf16_t* qkv, const int32_t* pos, const float* wq, const float* wk,
                               const float* cos_sin /*[max_seq][128][2]*/, float eps, float q_scale,
                               uint32_t M, uint32_t heads, uint32_t kv_heads, int k_only, hipStream_t st);

// vt[g][d][vt_start[seq] + pos] = v[token][g][d]  (keys contiguous per head dim: the A operand of
// O^T = V^T P^T); blk / seq_* / vt_start as for launch_attention.  Columns between a sequence's end and
// the next multiple of 32 are written as zeros.
hipError_t launch_v_transpose(const bf16_t* qkv, bf16_t* vt, const int32_t* blk, uint32_t nblk,
                              const int32_t* seq_start, const int32_t* seq_len, const int32_t* vt_start, uint32_t heads,
                              uint32_t kv_heads, uint32_t vt_ld, hipStream_t st);

// launch_qk_norm_rope(k_only = 1) + (with_vt != 0) launch_v_transpose in one launch: what the forward runs between
// the QKV GEMM and the attention.  with_vt = attention_reads_vt(...): the full-batch attention kernel reads V rows as
// they lie in qkv (transposed LDS reads), only the small-batch kernel wants V^T.
hipError_t launch_kv_prep(bf16_t* qkv, bf16_t* vt, const int32_t* pos, const float* wq, const float* wk,
                          const float* cos_sin, float eps, float q_scale, uint32_t M, uint32_t heads, uint32_t kv_heads,
                          const int32_t* blk, uint32_t nblk, const int32_t* seq_start, const int32_t* seq_len,
                          const int32_t* vt_start, uint32_t vt_ld, int with_vt, hipStream_t st);
bool attention_reads_vt(uint32_t nblk, uint32_t heads, uint32_t kv_heads);

// Bidirectional (optionally windowed) attention over packed sequences.
// blk[i] = {sequence, 128-row query super-block}; seq_start/seq_len in packed tokens; vt_start = first
// V^T column of the sequence (multiple of 32).  out [M, heads*256] bf16.
hipError_t launch_attention(const bf16_t* qkv, const bf16_t* vt, bf16_t* out, const int32_t* blk /*[nblk][2]*/,
                            uint32_t nblk, const int32_t* seq_start, const int32_t* seq_len,
                            const int32_t* vt_start, uint32_t vt_ld, uint32_t heads, uint32_t kv_heads,
                            uint32_t window /*0 = full*/, const float* q_norm_w /*nullable: q already normalised + rotated*/,
                            const float* cos_sin, float eps, float q_scale, hipStream_t st);

// pooled[b] = bf16(mean over the sequence's tokens of hidden[m]) (masked mean pool)
hipError_t launch_mean_pool(const float* hidden, const int32_t* seq_start, const int32_t* seq_len, bf16_t* pooled,
                            uint32_t B, uint32_t H, hipStream_t st);

// ---- the search-time forward (query_kernels.hip): ONE sequence of <= 64 tokens, 5 launches per layer + 2 for the head ----
struct QueryFwdLayer {
    const bf16_t *wqkv, *wo, *wgu, *wd;
    const float *n_in, *n_post_attn, *n_pre_ffw, *n_post_ffw, *n_q, *n_k;
};
struct QueryFwd {
    const int32_t* tok;               // device: token ids [T]
    const int32_t* pos;               // device: [kQueryFwdMaxTokens] = 0, 1, 2, ... (queries over 64 tokens: positions of the q / k rotation launch)
    uint32_t T;                       // tokens, 1..128: a launch parameter (the engine keeps one captured hipGraph per length)
    const bf16_t* emb;
    float embed_scale;
    const QueryFwdLayer* layer;       // host array [layers]
    uint32_t layers;
    const float* n_final;
    const bf16_t *dense1, *dense2;
    const float *rope_global, *rope_local;
    uint32_t hidden, heads, kv_heads, inter, dense_hidden, window, sliding_pattern;
    float eps, q_scale;
    // scratch, all for kQueryFwdMaxTokens rows: x0 / x1 [rows, hidden] f32 (the residual stream alternates), qkv [64, (heads + 2 kv) 256],
    // attn [64, heads 256], y [64, hidden], h [64, inter] bf16, d1 [dense_hidden] bf16, out [hidden] f32
    float *x0, *x1;
    bf16_t *qkv, *attn, *y, *h, *d1;
    float* out;
    unsigned long long* dbg;          // nullable: [5 layers + 2 kernel slots][256 workgroups][8] diagnostic stamps (query_kernels.hip)
};
constexpr uint32_t kQueryFwdMaxTokens = 128;
bool query_forward_supported(const EmbedGeom& g);
uint32_t query_forward_max_tokens(const EmbedGeom& g);            // 128 or 64: the longest sequence the chain serves for this geometry
hipError_t launch_query_forward(const QueryFwd& f, hipStream_t st);

// f32 -> bf16 (round to nearest even), n elements
hipError_t launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t st);

// C[M, N] = act(A W^T + bias) for M <= 64 rows through the search-time GEMM kernels (query_kernels.hip): N % 8 == 0,
// K % 128 == 0, out in {BF16, BF16_GELU, F32}; hipErrorNotSupported otherwise (the caller falls back to launch_gemm_bias's
// other kernels).  Not bit-identical to them: K is split over a workgroup's four waves.
hipError_t launch_gemm_small_rows(const bf16_t* A, const bf16_t* W, const float* bias, void* C, uint32_t M, uint32_t N,
                                  uint32_t K, uint32_t ldc, GemmOut out, hipStream_t st);

// launch_gemm_small_rows with the producer's residual add + LayerNorm in the prologue (BERT post-LN layers, M <= 64):
// x_out = bf16(LayerNorm(x + y) gamma + beta); C = act(x_out W^T + bias).  H in {256, 768, 1024}, N % 16 == 0, x_out != x.
hipError_t launch_gemm_small_rows_addln(const bf16_t* x, const bf16_t* y, const float* gamma, const float* beta, float eps,
                                        bf16_t* x_out, const bf16_t* W, const float* bias, void* C, uint32_t M, uint32_t N,
                                        uint32_t H, uint32_t ldc, GemmOut out, hipStream_t st);

}  // namespace cqs
