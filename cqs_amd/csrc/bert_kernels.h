// bert_kernels.h — launch interface of the BERT-family kernels (SPLADE encoder / cross-encoder reranker).
// Internal to libcqs_hip.so (public boundary: include/cqs_hip.h, BERT section).
#pragma once
#include "embed_kernels.h"

namespace cqs {

// out[m] = bf16(LayerNorm(word[tok[m]] + position[pos[m]] + token_type[tt[m]]) * gamma + beta); H % 128 == 0, H <= 1024
hipError_t launch_bert_embed_ln(const int32_t* tok, const int32_t* pos, const int32_t* tt, const bf16_t* word,
                                const bf16_t* posw, const bf16_t* typew, const float* gamma, const float* beta, float eps,
                                bf16_t* out, uint32_t M, uint32_t H, hipStream_t st);

// out[m] = bf16(LayerNorm(a[m] + r[m]) * gamma + beta); r == NULL: LayerNorm(a[m]); out may alias a
hipError_t launch_bert_add_ln(const bf16_t* a, const bf16_t* r, const float* gamma, const float* beta, float eps,
                              bf16_t* out, uint32_t M, uint32_t H, hipStream_t st);

// Multi-head bidirectional attention over packed sequences.  qkv [M, 3 H] bf16 (q | k | v, heads contiguous, biases
// already added), out [M, H]; head_dim 32 or 64; softmax(q k^T / sqrt(head_dim)) v.  Sequences of up to 512 tokens
// (max_len = the batch's longest) take the resident-key kernel, one workgroup per (sequence, head) [part];
// CQS_HIP_BERT_ATTN_RESIDENT=0 or a longer sequence takes the first-generation kernel, which walks
// blk[i] = {sequence, 64-query block}.
hipError_t launch_bert_attention(const bf16_t* qkv, bf16_t* out, const int32_t* blk /*[nblk][2]*/, uint32_t nblk,
                                 const int32_t* seq_start, const int32_t* seq_len, uint32_t B, uint32_t max_len,
                                 uint32_t heads, uint32_t head_dim, hipStream_t st);

// out[b] = mean over the sequence's tokens of x (mode 0; empty sequence: zeros) or its first token (mode 1), f32 [B, H]
hipError_t launch_bert_pool(const bf16_t* x, const int32_t* seq_start, const int32_t* seq_len, float* out, uint32_t B,
                            uint32_t H, int mode, hipStream_t st);

// x[i] <- ln(1 + x[i]) in place (the activation of src/splade/mod.rs:1049-1053 over launch_gemm_rowmax's maxima)
hipError_t launch_splade_activate(float* x, size_t n, hipStream_t st);

// Row b of dense [B, V] -> its entries > threshold as (id, weight), ascending id, the first `cap` of them at
// out_ids / out_w [b * cap ..]; out_count[b] = how many passed (may exceed cap).
hipError_t launch_splade_sparsify(const float* dense, uint32_t B, uint32_t V, float threshold, uint32_t cap, uint32_t* out_ids,
                                  float* out_w, uint32_t* out_count, hipStream_t st);

}  // namespace cqs
