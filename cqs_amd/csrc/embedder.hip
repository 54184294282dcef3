// embedder.hip — host side of the EmbeddingGemma forward behind the C ABI (include/cqs_hip.h,
// embed section).  Plays the role of the ORT `Session` inside the reference's `Embedder`
// (src/embedder/core.rs:34-226, `session.run` at :1097): one engine per device, inference
// serialised behind a mutex exactly like `Mutex<Option<Session>>` (core.rs:35-39), errors
// reported as status codes (-> `EmbedderError::InferenceFailed`, src/embedder/mod.rs:36-60).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "../../include/cqs_hip.h"
#include "abi_guard.h"
#include "roctx.h"
#include "embed_kernels.h"
#include "onnx_reader.h"
#include "safetensors_reader.h"

using cqs::bf16_t;

namespace {

uint16_t f32_to_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x0040u);  // keep NaN a NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
struct LayerW {
    bf16_t *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wd = nullptr;
    bf16_t *wo_p = nullptr, *wd_p = nullptr;      // wo / wd in the block order of the pair-split fused kernel (made at finalize)
    bf16_t* wgu_f = nullptr;                      // wgu rows interleaved per 4 (the 256-row kernel's in-register GeGLU pairing; made at finalize)
    bf16_t* wqkv_f = nullptr;                     // wqkv rows in tile order for the fused norm + RoPE epilogue (made at finalize)
    float *n_in = nullptr, *n_post_attn = nullptr, *n_pre_ffw = nullptr, *n_post_ffw = nullptr, *n_q = nullptr, *n_k = nullptr;
};

}  // namespace

struct cqs_hip_embedder {
    int device = 0;
    cqs_hip_embed_config cfg{};
    cqs::EmbedGeom g{};
    float last_ms = -1.f;

    bf16_t* emb = nullptr;
    float* n_final = nullptr;
    bf16_t *dense1 = nullptr, *dense2 = nullptr;
    std::vector<LayerW> L;
    std::vector<cqs::QueryFwdLayer> QL;   // the same pointers in the query path's layout (filled by finalize)
    bool query_path = false;              // geometry supported and not disabled by CQS_HIP_QUERY_PATH=0
    uint32_t query_max_tokens = 64;       // longest single sequence the search-time chain serves (cqs::query_forward_max_tokens)
    bool query_graph = true;              // CQS_HIP_QUERY_GRAPH=0: launch the query chain eagerly
    bool query_direct = true;             // CQS_HIP_QUERY_DIRECT=0: token ids / result always through copy calls
    bool single_ctx = false;              // CQS_HIP_EMBED_CONTEXTS=1: one execution context (A/B hook for the two-chain overlap)
    // o_proj / down fused with the residual add + both norms (gemm_rowfuse.hip).  Read ONCE, at finalize
    // (CQS_HIP_GEMM_FUSE_NORM / _MIN_ROWS); tests flip it through cqs_hip_debug_embedder_set_fuse_norm.
    uint32_t fuse_fallbacks = 0;          // times a pair-exchange timeout sent this engine back to the two-launch chain (0 or 1)
    int fuse_norm = 2;                    // 0 = two launches, 1 = the 64-row kernel of round 3, 2 = the pair-split kernel (128 rows x 384 columns)
    uint32_t fuse_min_rows = 4096;        // token count from which the fused kernel runs (pair-split kernel, ragged 10k-token batches, tickets in
                                          // flight: 9.6 k chunks/s at 1024-4096, 9.4 k at 2048 / 8192, 8.3 k at 12288; round 3's 64-row kernel needed 12288)
    bool geglu4 = true;                   // CQS_HIP_GEGLU4=0: the 256-row kernel's GeGLU pairs gate / up through an f32 LDS stage (round 2) instead of in registers
    bool fuse_qkv = true;                 // CQS_HIP_QKV_FUSE=0: QKV GEMM + kv_prep + the attention kernel's own Q norm instead of the fused epilogue
    unsigned* fuse_err = nullptr;         // pinned host word (device-visible at fuse_err_dev): set by the pair kernel if an exchange timed out
    unsigned* fuse_err_dev = nullptr;
    float *rope_global = nullptr, *rope_local = nullptr;  // [max_seq][128][2]
    std::map<std::string, bool> seen;
    bool finalized = false;

    // Execution contexts: a HIP stream + the activation scratch of one batch.  Consecutive tickets alternate between
    // the two, so the kernels of batch i+1 fill the CUs that batch i's kernels leave idle at their heads and tails
    // (every kernel here is one round of one workgroup per CU: load burst, compute, store burst, all CUs in phase;
    // measured +3-5 % forward throughput with two chains in flight).  Weights are shared.
    struct Ctx {
        hipStream_t stream = nullptr;
        // scratch, sized for tok_cap packed tokens / seq_cap sequences
        uint32_t tok_cap = 0, seq_cap = 0, vt_ld = 0, blk_cap = 0;
        float *x = nullptr, *hidden = nullptr, *out = nullptr;
        bf16_t *y = nullptr, *xn = nullptr, *qkv = nullptr, *vt = nullptr, *attn = nullptr, *h = nullptr, *pooled = nullptr, *d1 = nullptr;
        // per-batch integer tables in ONE device block (one H2D per batch from the slot's pinned twin):
        // [tok M][pos M][seq_start B][seq_len B][vt_start B][blk 2 nblk]; the pointers are carved per batch
        int32_t* d_meta = nullptr;
        size_t meta_cap = 0;   // int32 elements
        void* xch = nullptr;   // pair-split fused projection: the partners' row-sum granules (gemm_addnorm_pair_scratch_bytes(tok_cap))
        uint32_t xch_tag = 0;  // sequence number of the last launch that used xch (the granules' tag; never 0)
        int32_t *d_tok = nullptr, *d_pos = nullptr, *d_seq_start = nullptr, *d_seq_len = nullptr,
                *d_vt_start = nullptr, *d_blk = nullptr;
        // search-time path (query_kernels.hip): fixed 64-row scratch, allocated once and never moved, so that the
        // captured graphs' kernel arguments stay valid; the token ids arrive in q_meta by one H2D per query
        int32_t* q_meta = nullptr;
        int32_t* q_pos = nullptr;          // [kQueryFwdMaxTokens] = 0, 1, 2, ...
        float *q_x0 = nullptr, *q_x1 = nullptr, *q_out = nullptr;
        bf16_t *q_qkv = nullptr, *q_attn = nullptr, *q_y = nullptr, *q_h = nullptr, *q_d1 = nullptr;
        // one captured chain per query length T = 1..64 (T is a launch parameter of every kernel: no load waits for a
        // length read from memory, no row past T is touched), captured the first time a length is seen
        // [variant][T - 1]; variant 1 = "direct": the token ids are read from, and the sentence vector is written to,
        // the context's own pinned host buffers (q_tok_pin / q_out_pin, device-visible) - no H2D / D2H copy node and no
        // copy call around the chain.  Used when the context has no other ticket in flight (the blocking `embed_query`
        // call); a second ticket queued on the same context would overwrite those buffers, so it takes variant 0.
        hipGraph_t q_graph[2][cqs::kQueryFwdMaxTokens] = {};
        hipGraphExec_t q_exec[2][cqs::kQueryFwdMaxTokens] = {};
        bool q_capture_failed[2][cqs::kQueryFwdMaxTokens] = {};   // capture / instantiate refused for THIS (variant, length): it stays eager
        int32_t* q_tok_pin = nullptr;      // pinned host, 64 ids
        float* q_out_pin = nullptr;        // pinned host, [hidden]
        int32_t* q_tok_pin_dev = nullptr;  // their device addresses
        float* q_out_pin_dev = nullptr;
        unsigned long long* q_dbg = nullptr;   // CQS_HIP_QUERY_STAMPS=1: per-kernel, per-workgroup stamps of the last query
    };
    static constexpr int kCtx = 2;
    Ctx ctx[kCtx];
    // Search-time chain bookkeeping (cqs_hip_embedder_query_graph_stats).  A length's kernels differ by length class
    // (row-block variants), and their dynamic-LDS attributes are set on first launch - which must not happen inside a
    // stream capture: the first query of every length runs eagerly, the second is captured.
    bool q_len_ran[cqs::kQueryFwdMaxTokens] = {};   // this length's kernels have been launched once (attributes set)
    uint64_t q_captured = 0, q_capture_failures = 0, q_replays = 0, q_eager = 0;

    // Submission slots (pinned host staging + events): batch i+1 is packed and enqueued while batch i computes;
    // results come back through the slot's pinned `out` (cqs_hip_embed_submit / _collect).
    struct Slot {
        int32_t* meta = nullptr;   // pinned, same layout as d_meta
        size_t meta_cap = 0;
        float* out = nullptr;      // pinned [B, hidden]
        size_t out_cap = 0;        // floats
        uint32_t B = 0, M = 0, vt_cols = 0, nblk = 0;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;   // forward start / end (timing), ev_done after the D2H
        hipEvent_t done = nullptr;
        uint64_t ticket = 0;       // 0 = free
        int ctx = 0;               // execution context the ticket runs on
        bool direct = false;       // the result is in the context's q_out_pin, not in `out`
        bool collecting = false;   // a collect is waiting on this ticket: a second collector of the same ticket is refused
        bool batch_chain = false;  // the ticket ran the batch path (whose fused projections exchange row sums across workgroups)
        bool rerun = false;        // a pair exchange timed out while this ticket was in flight: recompute it at its collect
    };
    static constexpr int kSlots = 3;
    Slot slot[kSlots];
    uint64_t next_ticket = 1;
    int last_ctx = 1;              // context of the previous ticket (ties alternate)

    mutable std::mutex mu;
    std::atomic<bool> poisoned{false};
    std::string last_error;
};

namespace {

int32_t efail(cqs_hip_embedder* e, int32_t code, const std::string& what, hipError_t he = hipSuccess) {
    std::string msg = what;
    if (he != hipSuccess) msg += std::string(": ") + hipGetErrorString(he);
    if (e) {
        e->last_error = msg;
        if (code == CQS_HIP_ERR_DEVICE) e->poisoned.store(true, std::memory_order_release);
    }
    return code;
}
#define E_TRY(e, expr)                                                                                  \
    do {                                                                                                \
        hipError_t _h = (expr);                                                                         \
        if (_h != hipSuccess) return efail((e), _h == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE, #expr, _h); \
    } while (0)

template <class T>
hipError_t dmalloc(T** p, size_t count) { return hipMalloc((void**)p, count * sizeof(T)); }

uint32_t nqkv(const cqs::EmbedGeom& g) { return (g.heads + 2u * g.kv_heads) * g.head_dim; }

int32_t upload_bf16(cqs_hip_embedder* e, bf16_t* dst, const float* src, size_t count) {
    std::vector<uint16_t> tmp(count);
    for (size_t i = 0; i < count; ++i) tmp[i] = f32_to_bf16_bits(src[i]);
    E_TRY(e, hipMemcpy(dst, tmp.data(), count * 2, hipMemcpyHostToDevice));
    return CQS_HIP_OK;
}
int32_t upload_f32(cqs_hip_embedder* e, float* dst, const float* src, size_t count) {
    E_TRY(e, hipMemcpy(dst, src, count * 4, hipMemcpyHostToDevice));
    return CQS_HIP_OK;
}

using Ctx = cqs_hip_embedder::Ctx;

void free_query_scratch(Ctx& c) {
    for (int v = 0; v < 2; ++v)
        for (uint32_t i = 0; i < cqs::kQueryFwdMaxTokens; ++i) {
            if (c.q_exec[v][i]) (void)hipGraphExecDestroy(c.q_exec[v][i]);
            if (c.q_graph[v][i]) (void)hipGraphDestroy(c.q_graph[v][i]);
            c.q_exec[v][i] = nullptr; c.q_graph[v][i] = nullptr;
        }
    if (c.q_tok_pin) (void)hipHostFree(c.q_tok_pin);
    if (c.q_out_pin) (void)hipHostFree(c.q_out_pin);
    c.q_tok_pin = nullptr; c.q_out_pin = nullptr; c.q_tok_pin_dev = nullptr; c.q_out_pin_dev = nullptr;
    void** all[] = {(void**)&c.q_dbg, (void**)&c.q_pos, (void**)&c.q_meta, (void**)&c.q_x0, (void**)&c.q_x1, (void**)&c.q_out, (void**)&c.q_qkv, (void**)&c.q_attn,
                    (void**)&c.q_y, (void**)&c.q_h, (void**)&c.q_d1};
    for (void** p : all) { (void)hipFree(*p); *p = nullptr; }
}

void free_scratch(Ctx& c) {
    void** all[] = {(void**)&c.x, (void**)&c.y, (void**)&c.hidden, (void**)&c.out, (void**)&c.xn, (void**)&c.qkv,
                    (void**)&c.vt, (void**)&c.attn, (void**)&c.h, (void**)&c.pooled, (void**)&c.d1, (void**)&c.d_meta, &c.xch};
    for (void** p : all) { (void)hipFree(*p); *p = nullptr; }
    c.d_tok = c.d_pos = c.d_seq_start = c.d_seq_len = c.d_vt_start = c.d_blk = nullptr;
    c.tok_cap = c.seq_cap = c.vt_ld = c.blk_cap = 0;
    c.meta_cap = 0;
}

int32_t ensure_scratch(cqs_hip_embedder* e, Ctx& c, uint32_t M, uint32_t B, uint32_t vt_cols, uint32_t nblk) {
    const cqs::EmbedGeom& g = e->g;
    if (M <= c.tok_cap && B <= c.seq_cap && vt_cols <= c.vt_ld && nblk <= c.blk_cap) return CQS_HIP_OK;
    E_TRY(e, hipStreamSynchronize(c.stream));
    const uint32_t Mc = std::max(M, c.tok_cap), Bc = std::max(B, c.seq_cap);
    const uint32_t vc = std::max(vt_cols, c.vt_ld), bc = std::max(nblk, c.blk_cap);
    free_scratch(c);   // pointers nulled, capacities zeroed: a failed hipMalloc below leaves a consistent (empty) scratch
    const size_t H = g.hidden;
    E_TRY(e, dmalloc(&c.x, (size_t)Mc * H));
    E_TRY(e, dmalloc(&c.y, (size_t)Mc * H));
    E_TRY(e, dmalloc(&c.hidden, (size_t)Mc * H));
    E_TRY(e, dmalloc(&c.out, (size_t)Bc * H));
    E_TRY(e, dmalloc(&c.xn, (size_t)Mc * H));
    E_TRY(e, dmalloc(&c.qkv, (size_t)Mc * nqkv(g)));
    E_TRY(e, dmalloc(&c.vt, (size_t)g.kv_heads * g.head_dim * vc));
    E_TRY(e, hipMemset(c.vt, 0, (size_t)g.kv_heads * g.head_dim * vc * sizeof(bf16_t)));  // pad columns stay finite
    E_TRY(e, dmalloc(&c.attn, (size_t)Mc * g.heads * g.head_dim));
    E_TRY(e, dmalloc(&c.h, (size_t)Mc * g.inter));
    E_TRY(e, dmalloc(&c.pooled, (size_t)Bc * H));
    E_TRY(e, dmalloc(&c.d1, (size_t)Bc * g.dense_hidden));
    c.meta_cap = (size_t)2 * Mc + (size_t)3 * Bc + (size_t)2 * bc;
    E_TRY(e, dmalloc(&c.d_meta, c.meta_cap));
    E_TRY(e, hipMalloc(&c.xch, cqs::gemm_addnorm_pair_scratch_bytes(Mc)));
    E_TRY(e, hipMemset(c.xch, 0, cqs::gemm_addnorm_pair_scratch_bytes(Mc)));     // tag 0 = never written
    c.xch_tag = 0;
    c.tok_cap = Mc; c.seq_cap = Bc; c.vt_ld = vc; c.blk_cap = bc;
    return CQS_HIP_OK;
}

// ---- batch packing into a submission slot ------------------------------------------------------------------
using Slot = cqs_hip_embedder::Slot;

int32_t slot_reserve(cqs_hip_embedder* e, Slot& sl, uint32_t B, uint32_t M, uint32_t nblk) {
    const size_t need = (size_t)2 * M + (size_t)3 * B + (size_t)2 * nblk;
    if (need > sl.meta_cap) {
        if (sl.meta) (void)hipHostFree(sl.meta);
        sl.meta = nullptr; sl.meta_cap = 0;
        const size_t cap = need + need / 4 + 64;
        E_TRY(e, hipHostMalloc((void**)&sl.meta, cap * sizeof(int32_t), hipHostMallocDefault));
        sl.meta_cap = cap;
    }
    const size_t out_need = (size_t)B * e->g.hidden;
    if (out_need > sl.out_cap) {
        if (sl.out) (void)hipHostFree(sl.out);
        sl.out = nullptr; sl.out_cap = 0;
        E_TRY(e, hipHostMalloc((void**)&sl.out, (out_need + out_need / 4 + 64) * sizeof(float), hipHostMallocDefault));
        sl.out_cap = out_need + out_need / 4 + 64;
    }
    if (!sl.ev0) {
        E_TRY(e, hipEventCreate(&sl.ev0));
        E_TRY(e, hipEventCreate(&sl.ev1));
        E_TRY(e, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    sl.B = B; sl.M = M; sl.nblk = nblk;
    return CQS_HIP_OK;
}

// Fill the slot's tables from per-sequence lengths + a token fetcher; lens already validated (<= max_seq).
template <class TokAt>
int32_t slot_fill(cqs_hip_embedder* e, Slot& sl, const uint32_t* lens, uint32_t B, TokAt tok_at) {
    uint64_t M64 = 0, nblk64 = 0;
    for (uint32_t b = 0; b < B; ++b) { M64 += lens[b]; nblk64 += (lens[b] + 127u) / 128u; }
    if (M64 > 0x7FFFFFFFull) return efail(e, CQS_HIP_ERR_INVALID, "embed: batch holds too many tokens");
    int32_t rc = slot_reserve(e, sl, B, (uint32_t)M64, (uint32_t)nblk64);
    if (rc != CQS_HIP_OK) return rc;
    const uint32_t M = sl.M;
    int32_t* tok = sl.meta;
    int32_t* pos = tok + M;
    int32_t* seq_start = pos + M;
    int32_t* seq_len = seq_start + B;
    int32_t* vt_start = seq_len + B;
    int32_t* blk = vt_start + B;
    uint32_t m = 0, vcols = 0, nb = 0;
    const int64_t vocab = (int64_t)e->g.vocab;
    for (uint32_t b = 0; b < B; ++b) {
        const uint32_t len = lens[b];
        seq_start[b] = (int32_t)m; seq_len[b] = (int32_t)len; vt_start[b] = (int32_t)vcols;
        for (uint32_t j = 0; j < len; ++j) {
            const int64_t id = tok_at(b, j);
            if (id < 0 || id >= vocab) return efail(e, CQS_HIP_ERR_INVALID, "embed: token id out of range");
            tok[m + j] = (int32_t)id;
            pos[m + j] = (int32_t)j;
        }
        for (uint32_t sb = 0; sb * 128u < len; ++sb) { blk[2 * nb] = (int32_t)b; blk[2 * nb + 1] = (int32_t)sb; ++nb; }
        m += len;
        vcols += (len + 31u) / 32u * 32u;
    }
    sl.vt_cols = vcols + 32u;
    return CQS_HIP_OK;
}

// The padded [B, L] contract of `session.run` (src/embedder/core.rs:1031-1035): mask rows are 1...10...0.
int32_t pack_padded(cqs_hip_embedder* e, Slot& sl, const int64_t* ids, const int64_t* mask, uint32_t B, uint32_t L) {
    std::vector<uint32_t> lens(B);
    for (uint32_t b = 0; b < B; ++b) {
        uint32_t len = 0;
        while (len < L && mask[(size_t)b * L + len] != 0) ++len;
        for (uint32_t j = len; j < L; ++j)
            if (mask[(size_t)b * L + j] != 0)
                return efail(e, CQS_HIP_ERR_INVALID, "embed: attention_mask must be a right-padded prefix mask");
        if (len > e->g.max_seq) return efail(e, CQS_HIP_ERR_INVALID, "embed: sequence longer than max_seq");
        lens[b] = len;
    }
    return slot_fill(e, sl, lens.data(), B, [&](uint32_t b, uint32_t j) { return ids[(size_t)b * L + j]; });
}

int32_t pack_ragged(cqs_hip_embedder* e, Slot& sl, const int32_t* tokens, const uint32_t* lens, uint32_t B) {
    std::vector<size_t> start(B);
    size_t off = 0;
    for (uint32_t b = 0; b < B; ++b) {
        if (lens[b] > e->g.max_seq) return efail(e, CQS_HIP_ERR_INVALID, "embed: sequence longer than max_seq");
        start[b] = off;
        off += lens[b];
    }
    return slot_fill(e, sl, lens, B, [&](uint32_t b, uint32_t j) { return (int64_t)tokens[start[b] + j]; });
}

// Enqueue on the context's stream: tables H2D (one copy), the layers; leaves `hidden` (final norm, packed) in its scratch.
int32_t run_layers(cqs_hip_embedder* e, Ctx& c, Slot& sl) {
    const cqs::EmbedGeom& g = e->g;
    const uint32_t M = sl.M, H = g.hidden, B = sl.B;
    hipStream_t st = c.stream;
    int32_t rc = ensure_scratch(e, c, M, B, sl.vt_cols, sl.nblk);
    if (rc != CQS_HIP_OK) return rc;
    c.d_tok = c.d_meta;
    c.d_pos = c.d_tok + M;
    c.d_seq_start = c.d_pos + M;
    c.d_seq_len = c.d_seq_start + B;
    c.d_vt_start = c.d_seq_len + B;
    c.d_blk = c.d_vt_start + B;
    const size_t words = (size_t)2 * M + (size_t)3 * B + (size_t)2 * sl.nblk;
    E_TRY(e, hipMemcpyAsync(c.d_meta, sl.meta, words * sizeof(int32_t), hipMemcpyHostToDevice, st));
    E_TRY(e, hipEventRecord(sl.ev0, st));
    const uint32_t nblk = sl.nblk;
    const bool need_vt = cqs::attention_reads_vt(nblk, g.heads, g.kv_heads);
    int qkv_tn = (e->fuse_qkv && !need_vt) ? cqs::gemm_qkv_rope_tile(M, H, g.heads, g.kv_heads, g.head_dim) : 0;
    if (qkv_tn == 5 && !e->L[0].wqkv_f) qkv_tn = 0;
    const bool qkv_fused = qkv_tn != 0;
    E_TRY(e, cqs::launch_embed_norm(c.d_tok, e->emb, sqrtf((float)H), e->L[0].n_in, g.rms_eps, c.x, c.xn, M, H, st));
    for (uint32_t l = 0; l < g.layers; ++l) {
        const LayerW& w = e->L[l];
        const bool full = ((l + 1u) % g.sliding_pattern) == 0u;
        const float* rope = full ? e->rope_global : e->rope_local;
        // QKV projection.  Full rounds of 256 x 320 tiles: each tile = one q / k head + a slice of v, and the GEMM's epilogue
        // does the head's RMSNorm + RoPE (+ q scale) itself - no kv_prep launch, no Q prologue in the attention kernel
        if (qkv_fused) {
            cqs::QkvEpilogue ep{c.d_pos, w.n_q, w.n_k, rope, g.rms_eps, g.q_scale, g.heads, g.kv_heads};
            E_TRY(e, cqs::launch_gemm_qkv_rope(c.xn, qkv_tn == 5 ? w.wqkv_f : w.wqkv, c.qkv, M, H, qkv_tn, ep, st));
        } else {
            E_TRY(e, cqs::launch_gemm_bf16(c.xn, w.wqkv, c.qkv, M, nqkv(g), H, nqkv(g), cqs::GEMM_OUT_BF16, st));
            // k heads + V^T here; the attention kernel normalises / rotates its own Q fragments (q is 3/4 of the rope's bytes)
            E_TRY(e, cqs::launch_kv_prep(c.qkv, c.vt, c.d_pos, w.n_q, w.n_k, rope, g.rms_eps, g.q_scale, M, g.heads, g.kv_heads,
                                         c.d_blk, nblk, c.d_seq_start, c.d_seq_len, c.d_vt_start, c.vt_ld, need_vt ? 1 : 0, st));
        }
        E_TRY(e, cqs::launch_attention(c.qkv, c.vt, c.attn, c.d_blk, nblk, c.d_seq_start, c.d_seq_len,
                                       c.d_vt_start, c.vt_ld, g.heads, g.kv_heads, full ? 0u : g.window, qkv_fused ? nullptr : w.n_q, rope,
                                       g.rms_eps, g.q_scale, st));
        const int fuse = (e->fuse_norm && M >= e->fuse_min_rows && e->fuse_err_dev) ? e->fuse_norm : 0;
        auto proj_norm = [&](const bf16_t* Ain, const bf16_t* Wp, const bf16_t* Wpk, uint32_t Kp, const float* wpost, const float* wnext, float* outp, int fin) -> int32_t {
            if (fuse == 2 && Wpk && cqs::gemm_addnorm_supported(M, H, Kp)) {
                if (++c.xch_tag == 0u) c.xch_tag = 1u;
                E_TRY(e, cqs::launch_gemm_addnorm_pair(Ain, Wpk, c.x, wpost, wnext, g.rms_eps, c.xn, outp, fin, M, H, Kp, c.xch, c.xch_tag, e->fuse_err_dev, st));
            } else if (fuse == 1 && cqs::gemm_addnorm_supported(M, H, Kp)) {
                E_TRY(e, cqs::launch_gemm_addnorm(Ain, Wp, c.x, wpost, wnext, g.rms_eps, c.xn, outp, fin, M, H, Kp, st));
            } else {
                E_TRY(e, cqs::launch_gemm_bf16(Ain, Wp, c.y, M, H, Kp, H, cqs::GEMM_OUT_BF16, st));
                E_TRY(e, cqs::launch_add_norm(c.x, c.y, wpost, wnext, g.rms_eps, c.xn, outp, fin, M, H, st));
            }
            return CQS_HIP_OK;
        };
        // o_proj, then x += norm(y)(1 + w); xn = norm(x)(1 + w'): one launch where workgroups can own whole rows
        if ((rc = proj_norm(c.attn, w.wo, w.wo_p, g.heads * g.head_dim, w.n_post_attn, w.n_pre_ffw, nullptr, 0)) != CQS_HIP_OK) return rc;
        E_TRY(e, cqs::launch_gemm_bf16(c.xn, w.wgu, c.h, M, 2u * g.inter, H, g.inter, cqs::GEMM_OUT_GEGLU, st, nullptr, e->geglu4 ? w.wgu_f : nullptr));
        const bool last = (l + 1u == g.layers);
        const float* w_next = last ? e->n_final : e->L[l + 1].n_in;
        if ((rc = proj_norm(c.h, w.wd, w.wd_p, g.inter, w.n_post_ffw, w_next, c.hidden, last ? 1 : 0)) != CQS_HIP_OK) return rc;
    }
    return CQS_HIP_OK;
}

// ---- the search-time path: ONE sequence of <= 64 tokens (`embed_query`, src/embedder/core.rs:768-856) ---------------
// 98 launches (4 per layer + 2) of query_kernels.hip instead of the batch chain's ~230, replayed from a hipGraph
// captured on the context's second query (eager launches of 2-3 us kernels are host-bound: ~3.5 us of host time each).
bool slot_takes_query_path(const cqs_hip_embedder* e, const Slot& sl) {
    return e->query_path && sl.B == 1 && sl.M >= 1 && sl.M <= e->query_max_tokens;
}

int32_t ensure_query_scratch(cqs_hip_embedder* e, Ctx& c) {
    if (c.q_meta) return CQS_HIP_OK;
    const cqs::EmbedGeom& g = e->g;
    const size_t R = cqs::kQueryFwdMaxTokens, H = g.hidden;
    hipError_t he = hipSuccess;
    auto grab = [&](auto** p, size_t count) { if (he == hipSuccess) he = dmalloc(p, count); };
    grab(&c.q_meta, R + 1); grab(&c.q_pos, R); grab(&c.q_x0, R * H); grab(&c.q_x1, R * H); grab(&c.q_out, H);
    grab(&c.q_qkv, R * nqkv(g)); grab(&c.q_attn, R * g.heads * g.head_dim); grab(&c.q_y, R * H); grab(&c.q_h, R * g.inter);
    grab(&c.q_d1, (size_t)g.dense_hidden);
    // rows past a query's length are read (never used): keep them finite from the start
    if (he == hipSuccess && getenv("CQS_HIP_QUERY_STAMPS")) {
        const size_t words = ((size_t)g.layers * 5 + 2) * 256 * 8 * 2;   // x 2: CQS_HIP_QUERY_DEBUG_REPEAT=2
        grab(&c.q_dbg, words);
        if (he == hipSuccess) he = hipMemsetAsync(c.q_dbg, 0, words * 8, c.stream);
    }
    if (he == hipSuccess) he = hipHostMalloc((void**)&c.q_tok_pin, R * sizeof(int32_t), hipHostMallocDefault);
    if (he == hipSuccess) he = hipHostMalloc((void**)&c.q_out_pin, H * sizeof(float), hipHostMallocDefault);
    if (he == hipSuccess) he = hipHostGetDevicePointer((void**)&c.q_tok_pin_dev, c.q_tok_pin, 0);
    if (he == hipSuccess) he = hipHostGetDevicePointer((void**)&c.q_out_pin_dev, c.q_out_pin, 0);
    if (he == hipSuccess) memset(c.q_tok_pin, 0, R * sizeof(int32_t));
    if (he == hipSuccess) {
        std::vector<int32_t> iota(R);
        for (size_t i = 0; i < R; ++i) iota[i] = (int32_t)i;
        he = hipMemcpy(c.q_pos, iota.data(), R * sizeof(int32_t), hipMemcpyHostToDevice);
    }
    if (he == hipSuccess) he = hipMemsetAsync(c.q_x0, 0, R * H * 4, c.stream);
    if (he == hipSuccess) he = hipMemsetAsync(c.q_x1, 0, R * H * 4, c.stream);
    if (he == hipSuccess) he = hipMemsetAsync(c.q_qkv, 0, R * nqkv(g) * 2, c.stream);
    if (he == hipSuccess) he = hipMemsetAsync(c.q_attn, 0, R * g.heads * g.head_dim * 2, c.stream);
    if (he == hipSuccess) he = hipMemsetAsync(c.q_y, 0, R * H * 2, c.stream);
    if (he == hipSuccess) he = hipMemsetAsync(c.q_h, 0, R * g.inter * 2, c.stream);
    if (he != hipSuccess) {
        free_query_scratch(c);
        return efail(e, he == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE, "query scratch", he);
    }
    return CQS_HIP_OK;
}

// Enqueue on the context's stream: [T, tokens] H2D, the chain (graph replay once captured); leaves the sentence
// vector (f32 [hidden], not normalised) in c.q_out.
int32_t run_query(cqs_hip_embedder* e, Ctx& c, Slot& sl, bool direct) {
    const cqs::EmbedGeom& g = e->g;
    hipStream_t st = c.stream;
    int32_t rc = ensure_query_scratch(e, c);
    if (rc != CQS_HIP_OK) return rc;
    const int var = direct ? 1 : 0;
    if (direct) memcpy(c.q_tok_pin, sl.meta, (size_t)sl.M * sizeof(int32_t));     // the context is idle: nobody reads it now
    else E_TRY(e, hipMemcpyAsync(c.q_meta, sl.meta, (size_t)sl.M * sizeof(int32_t), hipMemcpyHostToDevice, st));   // slot_fill: the token ids come first
    E_TRY(e, hipEventRecord(sl.ev0, st));
    cqs::QueryFwd f{};
    f.tok = c.q_meta; f.pos = c.q_pos; f.T = sl.M; f.emb = e->emb; f.embed_scale = sqrtf((float)g.hidden); f.layer = e->QL.data(); f.layers = g.layers;
    f.n_final = e->n_final; f.dense1 = e->dense1; f.dense2 = e->dense2; f.rope_global = e->rope_global; f.rope_local = e->rope_local;
    f.hidden = g.hidden; f.heads = g.heads; f.kv_heads = g.kv_heads; f.inter = g.inter; f.dense_hidden = g.dense_hidden;
    f.window = g.window; f.sliding_pattern = g.sliding_pattern; f.eps = g.rms_eps; f.q_scale = g.q_scale;
    f.dbg = c.q_dbg;
    f.x0 = c.q_x0; f.x1 = c.q_x1; f.qkv = c.q_qkv; f.attn = c.q_attn; f.y = c.q_y; f.h = c.q_h; f.d1 = c.q_d1; f.out = c.q_out;
    if (direct) { f.tok = c.q_tok_pin_dev; f.out = c.q_out_pin_dev; }
    const int gi = (int)sl.M - 1;
    if (c.q_exec[var][gi]) {
        E_TRY(e, hipGraphLaunch(c.q_exec[var][gi], st));
        e->q_replays++;
        return CQS_HIP_OK;
    }
    if (e->query_graph && !c.q_capture_failed[var][gi] && e->q_len_ran[gi]) {
        // capture the chain (kernel launches only; every argument is a fixed device address), instantiate, replay
        const char* stage = "hipStreamBeginCapture";
        hipError_t he = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (he == hipSuccess) {
            const hipError_t le = cqs::launch_query_forward(f, st);
            hipGraph_t gr = nullptr;
            he = hipStreamEndCapture(st, &gr);
            stage = "hipStreamEndCapture";
            if (he == hipSuccess && le != hipSuccess) { he = le; stage = "a launch inside the capture"; }
            if (he == hipSuccess) { he = hipGraphInstantiate(&c.q_exec[var][gi], gr, nullptr, nullptr, 0); stage = "hipGraphInstantiate"; }
            if (he == hipSuccess) {
                c.q_graph[var][gi] = gr;
                e->q_captured++;
                E_TRY(e, hipGraphLaunch(c.q_exec[var][gi], st));
                e->q_replays++;
                return CQS_HIP_OK;
            }
            if (gr) (void)hipGraphDestroy(gr);
            c.q_exec[var][gi] = nullptr;
        }
        (void)hipGetLastError();
        // not a device failure (the eager chain below computes the same thing), but never silent: counted, latched for
        // this (variant, length) only, and the cause is left in last_error
        c.q_capture_failed[var][gi] = true;
        e->q_capture_failures++;
        e->last_error = std::string("query graph capture failed at ") + stage + " (" + std::to_string(sl.M) + " tokens, variant " +
                        std::to_string(var) + "): " + hipGetErrorString(he) + "; this length runs eagerly";
    }
    E_TRY(e, cqs::launch_query_forward(f, st));
    e->q_len_ran[gi] = true;
    e->q_eager++;
    return CQS_HIP_OK;
}

const char* kLayerTensors[] = {"input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight",
                               "self_attn.v_proj.weight", "self_attn.o_proj.weight", "self_attn.q_norm.weight",
                               "self_attn.k_norm.weight", "post_attention_layernorm.weight",
                               "pre_feedforward_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                               "mlp.down_proj.weight", "post_feedforward_layernorm.weight"};

// safetensors file -> set_tensor under the names `rename` maps to ("" = skip); reader: safetensors_reader.cpp
int32_t load_safetensors(cqs_hip_embedder* e, const std::string& path, const std::function<std::string(const std::string&)>& rename) {
    std::string err;
    int32_t inner = CQS_HIP_OK;
    const int fed = cqs_st::load(path, [&](const std::string& raw_name, const float* data, uint64_t count) -> int {
        const std::string name = rename(raw_name);
        if (name.empty()) return 0;
        inner = cqs_hip_embedder_set_tensor(e, name.c_str(), data, count);
        return inner == CQS_HIP_OK ? 1 : -1;
    }, err);
    if (fed < 0) return inner != CQS_HIP_OK ? inner : efail(e, CQS_HIP_ERR_INVALID, "load_dir: " + err);
    return CQS_HIP_OK;
}

// Is `name` one of the tensors cqs_hip_embedder_set_tensor accepts for this geometry?
bool known_tensor(const cqs_hip_embedder* e, const std::string& name) {
    if (name == "embed_tokens.weight" || name == "norm.weight" || name == "dense1.weight" || name == "dense2.weight") return true;
    if (name.rfind("layers.", 0) != 0) return false;
    const size_t dot = name.find('.', 7);
    if (dot == std::string::npos) return false;
    for (size_t i = 7; i < dot; ++i) if (name[i] < '0' || name[i] > '9') return false;
    if ((uint32_t)atoi(name.substr(7, dot - 7).c_str()) >= e->g.layers) return false;
    const std::string t = name.substr(dot + 1);
    for (const char* k : kLayerTensors) if (t == k) return true;
    return false;
}

}  // namespace

extern "C" {

void cqs_hip_embed_config_default(cqs_hip_embed_config* c) CQS_ABI_TRY {
    if (!c) return;
    c->vocab_size = 262144; c->hidden = 768; c->layers = 24; c->heads = 3; c->kv_heads = 1; c->head_dim = 256;
    c->intermediate = 1152; c->dense_hidden = 3072; c->sliding_window = 512; c->sliding_pattern = 6; c->max_seq = 2048;
    c->rms_eps = 1e-6f; c->rope_theta_global = 1e6f; c->rope_theta_local = 1e4f; c->query_pre_attn_scalar = 256.f;
} CQS_ABI_CATCH_VOID

int32_t cqs_hip_embedder_create(const cqs_hip_embed_config* c, int32_t device, cqs_hip_embedder** out) CQS_ABI_TRY {
    if (!c || !out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    if (c->head_dim != 256 || c->hidden == 0 || c->hidden % 256 || c->hidden > 1024 || c->intermediate % 64 ||
        c->dense_hidden % 128 || c->kv_heads == 0 || c->heads % c->kv_heads || c->heads / c->kv_heads > 4 || c->heads + c->kv_heads > 8 ||
        c->layers == 0 || c->sliding_pattern == 0 || c->max_seq == 0 || c->vocab_size == 0 || c->hidden % 128)
        return CQS_HIP_ERR_INVALID;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return CQS_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= cnt) return CQS_HIP_ERR_INVALID;
    cqs_hip_embedder* e = new (std::nothrow) cqs_hip_embedder();
    if (!e) return CQS_HIP_ERR_NOMEM;
    struct Owner {   // an exception below (L.resize, the rope table) must not leak the handle or its device buffers
        cqs_hip_embedder* p;
        ~Owner() { if (p) cqs_hip_embedder_destroy(p); }
    } owner{e};
    e->device = device;
    e->cfg = *c;
    cqs::EmbedGeom& g = e->g;
    g.vocab = c->vocab_size; g.hidden = c->hidden; g.layers = c->layers; g.heads = c->heads; g.kv_heads = c->kv_heads;
    g.head_dim = c->head_dim; g.inter = c->intermediate; g.dense_hidden = c->dense_hidden;
    g.window = c->sliding_window / 2u + 1u;  // bidirectional (configuration_gemma3.py:105-106)
    g.sliding_pattern = c->sliding_pattern; g.max_seq = c->max_seq; g.rms_eps = c->rms_eps;
    g.theta_global = c->rope_theta_global; g.theta_local = c->rope_theta_local;
    g.q_scale = 1.0f / sqrtf(c->query_pre_attn_scalar);
    bool ok = hipSetDevice(device) == hipSuccess;
    for (cqs_hip_embedder::Ctx& c : e->ctx) ok = ok && hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) == hipSuccess;
    if (ok && hipHostMalloc((void**)&e->fuse_err, 64, hipHostMallocDefault) == hipSuccess) {
        *e->fuse_err = 0u;
        if (hipHostGetDevicePointer((void**)&e->fuse_err_dev, e->fuse_err, 0) != hipSuccess) { (void)hipGetLastError(); e->fuse_err_dev = nullptr; }
    } else (void)hipGetLastError();                        // (no mappable word: the engine keeps the two-launch chain)
    const size_t H = g.hidden, D = g.head_dim;
    e->L.resize(g.layers);
    ok = ok && dmalloc(&e->emb, (size_t)g.vocab * H) == hipSuccess && dmalloc(&e->n_final, H) == hipSuccess &&
         dmalloc(&e->dense1, (size_t)g.dense_hidden * H) == hipSuccess && dmalloc(&e->dense2, H * g.dense_hidden) == hipSuccess &&
         dmalloc(&e->rope_global, (size_t)g.max_seq * 256) == hipSuccess && dmalloc(&e->rope_local, (size_t)g.max_seq * 256) == hipSuccess;
    for (uint32_t l = 0; ok && l < g.layers; ++l) {
        LayerW& w = e->L[l];
        ok = dmalloc(&w.wqkv, (size_t)nqkv(g) * H) == hipSuccess && dmalloc(&w.wo, H * g.heads * D) == hipSuccess &&
             dmalloc(&w.wgu, (size_t)2 * g.inter * H) == hipSuccess && dmalloc(&w.wd, H * g.inter) == hipSuccess &&
             dmalloc(&w.n_in, H) == hipSuccess && dmalloc(&w.n_post_attn, H) == hipSuccess && dmalloc(&w.n_pre_ffw, H) == hipSuccess &&
             dmalloc(&w.n_post_ffw, H) == hipSuccess && dmalloc(&w.n_q, D) == hipSuccess && dmalloc(&w.n_k, D) == hipSuccess;
    }
    if (!ok) return CQS_HIP_ERR_NOMEM;
    // RoPE tables, computed like transformers does (fp32 inv_freq, fp32 angle; modeling_gemma3.py:188-224)
    std::vector<float> tab((size_t)g.max_seq * 256);
    for (int t = 0; t < 2; ++t) {
        const float theta = t == 0 ? g.theta_global : g.theta_local;
        for (uint32_t pz = 0; pz < g.max_seq; ++pz)
            for (uint32_t i = 0; i < 128; ++i) {
                const float inv = 1.0f / powf(theta, (float)(2 * i) / 256.0f);
                const float a = (float)pz * inv;
                tab[((size_t)pz * 128 + i) * 2] = cosf(a);
                tab[((size_t)pz * 128 + i) * 2 + 1] = sinf(a);
            }
        if (hipMemcpy(t == 0 ? e->rope_global : e->rope_local, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return CQS_HIP_ERR_DEVICE;
    }
    owner.p = nullptr;
    *out = e;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

int32_t cqs_hip_embedder_set_tensor(cqs_hip_embedder* e, const char* cname, const float* data, uint64_t count) CQS_ABI_TRY {
    if (!e || !cname || !data) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->finalized) return efail(e, CQS_HIP_ERR_INVALID, "set_tensor after finalize");
    if (hipSetDevice(e->device) != hipSuccess) return efail(e, CQS_HIP_ERR_DEVICE, "hipSetDevice");
    const cqs::EmbedGeom& g = e->g;
    const std::string name(cname);
    const size_t H = g.hidden, D = g.head_dim, I = g.inter;
    auto need = [&](size_t n) { return count == n; };
    int32_t rc = CQS_HIP_ERR_INVALID;
    if (name == "embed_tokens.weight") { if (need((size_t)g.vocab * H)) rc = upload_bf16(e, e->emb, data, count); }
    else if (name == "norm.weight") { if (need(H)) rc = upload_f32(e, e->n_final, data, count); }
    else if (name == "dense1.weight") { if (need((size_t)g.dense_hidden * H)) rc = upload_bf16(e, e->dense1, data, count); }
    else if (name == "dense2.weight") { if (need((size_t)g.dense_hidden * H)) rc = upload_bf16(e, e->dense2, data, count); }
    else if (name.rfind("layers.", 0) == 0) {
        const size_t dot = name.find('.', 7);
        if (dot == std::string::npos) return efail(e, CQS_HIP_ERR_INVALID, "set_tensor: bad name " + name);
        const uint32_t l = (uint32_t)atoi(name.substr(7, dot - 7).c_str());
        if (l >= g.layers) return efail(e, CQS_HIP_ERR_INVALID, "set_tensor: layer out of range in " + name);
        LayerW& w = e->L[l];
        const std::string t = name.substr(dot + 1);
        if (t == "input_layernorm.weight") { if (need(H)) rc = upload_f32(e, w.n_in, data, count); }
        else if (t == "post_attention_layernorm.weight") { if (need(H)) rc = upload_f32(e, w.n_post_attn, data, count); }
        else if (t == "pre_feedforward_layernorm.weight") { if (need(H)) rc = upload_f32(e, w.n_pre_ffw, data, count); }
        else if (t == "post_feedforward_layernorm.weight") { if (need(H)) rc = upload_f32(e, w.n_post_ffw, data, count); }
        else if (t == "self_attn.q_norm.weight") { if (need(D)) rc = upload_f32(e, w.n_q, data, count); }
        else if (t == "self_attn.k_norm.weight") { if (need(D)) rc = upload_f32(e, w.n_k, data, count); }
        // q | k | v rows stacked into one [nqkv, H] matrix: one GEMM produces all three
        else if (t == "self_attn.q_proj.weight") { if (need(g.heads * D * H)) rc = upload_bf16(e, w.wqkv, data, count); }
        else if (t == "self_attn.k_proj.weight") { if (need(g.kv_heads * D * H)) rc = upload_bf16(e, w.wqkv + (size_t)g.heads * D * H, data, count); }
        else if (t == "self_attn.v_proj.weight") { if (need(g.kv_heads * D * H)) rc = upload_bf16(e, w.wqkv + (size_t)(g.heads + g.kv_heads) * D * H, data, count); }
        else if (t == "self_attn.o_proj.weight") { if (need(H * g.heads * D)) rc = upload_bf16(e, w.wo, data, count); }
        else if (t == "mlp.down_proj.weight") { if (need(H * I)) rc = upload_bf16(e, w.wd, data, count); }
        else if (t == "mlp.gate_proj.weight" || t == "mlp.up_proj.weight") {
            // gate / up rows interleaved per 32 channels: rows [64c, 64c+32) = gate channels [32c, 32c+32),
            // rows [64c+32, 64c+64) = the same channels' up rows -> the GEMM epilogue fuses gelu(gate)*up
            if (need(I * H)) {
                const size_t off = (t == "mlp.up_proj.weight") ? 32 : 0;
                rc = CQS_HIP_OK;
                for (size_t c = 0; c < I / 32 && rc == CQS_HIP_OK; ++c)
                    rc = upload_bf16(e, w.wgu + (64 * c + off) * H, data + 32 * c * H, 32 * H);
            }
        } else return efail(e, CQS_HIP_ERR_INVALID, "set_tensor: unknown tensor " + name);
    } else return efail(e, CQS_HIP_ERR_INVALID, "set_tensor: unknown tensor " + name);
    if (rc == CQS_HIP_ERR_INVALID) return efail(e, rc, "set_tensor: wrong element count for " + name);
    if (rc == CQS_HIP_OK) e->seen[name] = true;
    return rc;
} CQS_ABI_CATCH(e)

int32_t cqs_hip_embedder_finalize(cqs_hip_embedder* e) CQS_ABI_TRY {
    if (!e) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    std::vector<std::string> need = {"embed_tokens.weight", "norm.weight", "dense1.weight", "dense2.weight"};
    for (uint32_t l = 0; l < e->g.layers; ++l)
        for (const char* t : kLayerTensors) need.push_back("layers." + std::to_string(l) + "." + t);
    for (const std::string& n : need)
        if (!e->seen.count(n)) return efail(e, CQS_HIP_ERR_INVALID, "finalize: missing tensor " + n);
    // the pair-split fused projection kernel streams its weights in 2 KB blocks: re-order wo / wd once (hidden 768 only)
    if (e->g.hidden == 768u && (e->g.heads * e->g.head_dim) % 64u == 0u && e->g.inter % 64u == 0u) {
        if (hipSetDevice(e->device) != hipSuccess) return efail(e, CQS_HIP_ERR_DEVICE, "hipSetDevice");
        const uint32_t H = e->g.hidden, Ko = e->g.heads * e->g.head_dim, Kd = e->g.inter;
        for (LayerW& w : e->L) {
            if (!w.wo_p) E_TRY(e, dmalloc(&w.wo_p, (size_t)H * Ko));
            if (!w.wd_p) E_TRY(e, dmalloc(&w.wd_p, (size_t)H * Kd));
            E_TRY(e, cqs::launch_pack_rowfuse_w(w.wo, w.wo_p, H, Ko, nullptr));
            E_TRY(e, cqs::launch_pack_rowfuse_w(w.wd, w.wd_p, H, Kd, nullptr));
        }
        E_TRY(e, hipDeviceSynchronize());
    }
    if (e->g.heads == 3u * e->g.kv_heads && e->g.head_dim == 256u) {     // QKV rows in tile order for the fused norm + RoPE epilogue
        if (hipSetDevice(e->device) != hipSuccess) return efail(e, CQS_HIP_ERR_DEVICE, "hipSetDevice");
        for (LayerW& w : e->L) {
            if (!w.wqkv_f) E_TRY(e, dmalloc(&w.wqkv_f, (size_t)(e->g.heads + e->g.kv_heads) * 320u * e->g.hidden));
            E_TRY(e, cqs::launch_permute_qkv_rows(w.wqkv, w.wqkv_f, e->g.heads, e->g.kv_heads, e->g.hidden, nullptr));
        }
        E_TRY(e, hipDeviceSynchronize());
    }
    if (const char* fq = getenv("CQS_HIP_QKV_FUSE")) e->fuse_qkv = fq[0] != '0';
    if (const char* g4 = getenv("CQS_HIP_GEGLU4")) e->geglu4 = g4[0] != '0';
    if ((2u * e->g.inter) % 64u == 0u) {                               // gate / up rows interleaved per 4 for the in-register pairing
        if (hipSetDevice(e->device) != hipSuccess) return efail(e, CQS_HIP_ERR_DEVICE, "hipSetDevice");
        for (LayerW& w : e->L) {
            if (!w.wgu_f) E_TRY(e, dmalloc(&w.wgu_f, (size_t)2 * e->g.inter * e->g.hidden));
            E_TRY(e, cqs::launch_permute_geglu_rows(w.wgu, w.wgu_f, 2u * e->g.inter, e->g.hidden, nullptr));
        }
        E_TRY(e, hipDeviceSynchronize());
    }
    e->QL.resize(e->L.size());
    for (size_t l = 0; l < e->L.size(); ++l) {
        const LayerW& w = e->L[l];
        e->QL[l] = cqs::QueryFwdLayer{w.wqkv, w.wo, w.wgu, w.wd, w.n_in, w.n_post_attn, w.n_pre_ffw, w.n_post_ffw, w.n_q, w.n_k};
    }
    const char* qp = getenv("CQS_HIP_QUERY_PATH");
    const char* qg = getenv("CQS_HIP_QUERY_GRAPH");
    e->query_path = cqs::query_forward_supported(e->g) && !(qp && qp[0] == '0');
    e->query_graph = !(qg && qg[0] == '0');
    e->query_max_tokens = cqs::query_forward_max_tokens(e->g);
    const char* qd = getenv("CQS_HIP_QUERY_DIRECT");
    e->query_direct = !(qd && qd[0] == '0');
    const char* ec = getenv("CQS_HIP_EMBED_CONTEXTS");
    e->single_ctx = ec && ec[0] == '1';
    if (const char* fn = getenv("CQS_HIP_GEMM_FUSE_NORM")) e->fuse_norm = fn[0] == '0' ? 0 : (fn[0] == '1' ? 1 : 2);
    if (const char* fr = getenv("CQS_HIP_GEMM_FUSE_NORM_MIN_ROWS")) e->fuse_min_rows = (uint32_t)atoi(fr);
    e->finalized = true;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

int32_t cqs_hip_embedder_load_dir(const char* dir, const cqs_hip_embed_config* cfg, int32_t device, cqs_hip_embedder** out) CQS_ABI_TRY {
    if (!dir || !out) return CQS_HIP_ERR_INVALID;
    cqs_hip_embed_config c;
    if (cfg) c = *cfg; else cqs_hip_embed_config_default(&c);
    cqs_hip_embedder* e = nullptr;
    int32_t rc = cqs_hip_embedder_create(&c, device, &e);
    if (rc != CQS_HIP_OK) return rc;
    struct Owner {   // every exit but the last, exceptions out of the readers included, gives the handle back
        cqs_hip_embedder* p;
        ~Owner() { if (p) cqs_hip_embedder_destroy(p); }
    } owner{e};
    const std::string d(dir);
    auto strip = [](const std::string& n) -> std::string {  // HF checkpoints prefix text-model tensors with "model."
        if (n.rfind("model.", 0) == 0) return n.substr(6);
        return n;
    };
    // What the reference's local-model hook holds (CQS_ONNX_DIR, src/embedder/download.rs:12-41): the structured
    // layout `onnx/model.onnx` (src/embedder/models.rs:455-457) or the flat `model.onnx`, each with its external-data
    // sidecar next to it (download.rs:82).  A Hugging Face checkpoint directory (model.safetensors + 2_Dense/ +
    // 3_Dense/) is read too.
    struct stat stt;
    std::string onnx;
    for (const char* cand : {"/onnx/model.onnx", "/model.onnx"})
        if (onnx.empty() && stat((d + cand).c_str(), &stt) == 0) onnx = d + cand;
    if (!onnx.empty()) {
        std::string err;
        const int fed = cqs_onnx::load(onnx, e->g.hidden, e->g.dense_hidden,
            [&](const std::string& name, const float* data, uint64_t count, const std::vector<uint64_t>&) -> int {
                if (!known_tensor(e, name)) return 0;                      // graph constants etc.
                return cqs_hip_embedder_set_tensor(e, name.c_str(), data, count) == CQS_HIP_OK ? 1 : -1;
            }, err);
        if (fed < 0) rc = efail(e, CQS_HIP_ERR_INVALID, "load_dir: " + err + (e->last_error.empty() ? "" : " (" + e->last_error + ")"));
    } else {
        rc = load_safetensors(e, d + "/model.safetensors", strip);
        if (rc == CQS_HIP_OK)
            rc = load_safetensors(e, d + "/2_Dense/model.safetensors", [](const std::string& n) { return n == "linear.weight" ? std::string("dense1.weight") : std::string(); });
        if (rc == CQS_HIP_OK)
            rc = load_safetensors(e, d + "/3_Dense/model.safetensors", [](const std::string& n) { return n == "linear.weight" ? std::string("dense2.weight") : std::string(); });
    }
    if (rc == CQS_HIP_OK) rc = cqs_hip_embedder_finalize(e);
    if (rc != CQS_HIP_OK) {
        fprintf(stderr, "[cqs_hip] embedder load_dir failed: %s\n", e->last_error.c_str());
        return rc;
    }
    owner.p = nullptr;
    *out = e;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

void cqs_hip_embedder_destroy(cqs_hip_embedder* e) CQS_ABI_TRY {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (Ctx& c : e->ctx)
        if (c.stream) (void)hipStreamSynchronize(c.stream);
    void* g[] = {e->emb, e->n_final, e->dense1, e->dense2, e->rope_global, e->rope_local};
    for (void* p : g) (void)hipFree(p);
    if (e->fuse_err) (void)hipHostFree(e->fuse_err);
    for (Ctx& c : e->ctx) { free_scratch(c); free_query_scratch(c); }
    for (cqs_hip_embedder::Slot& sl : e->slot) {
        if (sl.meta) (void)hipHostFree(sl.meta);
        if (sl.out) (void)hipHostFree(sl.out);
        if (sl.ev0) (void)hipEventDestroy(sl.ev0);
        if (sl.ev1) (void)hipEventDestroy(sl.ev1);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    for (LayerW& w : e->L) {
        void* ws[] = {w.wgu_f, w.wqkv_f, w.wo_p, w.wd_p, w.wqkv, w.wo, w.wgu, w.wd, w.n_in, w.n_post_attn, w.n_pre_ffw, w.n_post_ffw, w.n_q, w.n_k};
        for (void* p : ws) (void)hipFree(p);
    }
    for (Ctx& c : e->ctx)
        if (c.stream) (void)hipStreamDestroy(c.stream);
    delete e;
} CQS_ABI_CATCH_VOID

uint32_t cqs_hip_embedder_dim(const cqs_hip_embedder* e) CQS_ABI_TRY { return e ? e->g.hidden : 0; } CQS_ABI_CATCH_VAL(0)
uint32_t cqs_hip_embedder_max_seq(const cqs_hip_embedder* e) CQS_ABI_TRY { return e ? e->g.max_seq : 0; } CQS_ABI_CATCH_VAL(0)
int32_t cqs_hip_embedder_poisoned(const cqs_hip_embedder* e) CQS_ABI_TRY { return e && e->poisoned.load(std::memory_order_acquire) ? 1 : 0; } CQS_ABI_CATCH_NOHANDLE
float cqs_hip_embedder_last_ms(const cqs_hip_embedder* e) CQS_ABI_TRY { return e ? e->last_ms : -1.f; } CQS_ABI_CATCH_VAL(-1.f)
size_t cqs_hip_embedder_last_error(const cqs_hip_embedder* e, char* buf, size_t cap) CQS_ABI_TRY {
    if (!e || !buf || cap == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    const size_t m = e->last_error.size() < cap - 1 ? e->last_error.size() : cap - 1;
    memcpy(buf, e->last_error.data(), m);
    buf[m] = 0;
    return m;
} CQS_ABI_CATCH_VAL(0)

// ---- submit / collect ---------------------------------------------------------------------------------------
// submit: validate + pack into a free slot's pinned tables (host work), enqueue H2D + forward + pool / dense +
// D2H on the engine's stream, return a ticket; nothing waits for the device.  collect: wait for that slot's
// event, hand the rows out.  With kSlots tickets in flight the host packs batch i+1 (and the caller tokenises
// batch i+2) while the device runs batch i - the overlap the reference gets from its parse -> embed -> write
// channel pipeline (src/cli/pipeline/mod.rs:61-244), moved under the `session.run` seam.
namespace {

// The batch path of one packed slot: H2D of its tables, the layers, pool + Dense head, D2H, the slot's events.
int32_t enqueue_batch_chain(cqs_hip_embedder* e, Ctx& c, cqs_hip_embedder::Slot& sl) {
    hipStream_t st = c.stream;
    const uint32_t H = e->g.hidden, B = sl.B;
    int32_t rc = run_layers(e, c, sl);
    if (rc != CQS_HIP_OK) return rc;
    const cqs::EmbedGeom& g = e->g;
    E_TRY(e, cqs::launch_mean_pool(c.hidden, c.d_seq_start, c.d_seq_len, c.pooled, B, H, st));
    E_TRY(e, cqs::launch_gemm_skinny(c.pooled, e->dense1, c.d1, B, g.dense_hidden, H, g.dense_hidden, cqs::GEMM_OUT_BF16, st));
    E_TRY(e, cqs::launch_gemm_skinny(c.d1, e->dense2, c.out, B, H, g.dense_hidden, H, cqs::GEMM_OUT_F32, st));
    E_TRY(e, hipEventRecord(sl.ev1, st));
    E_TRY(e, hipMemcpyAsync(sl.out, c.out, (size_t)B * H * sizeof(float), hipMemcpyDeviceToHost, st));
    E_TRY(e, hipEventRecord(sl.done, st));
    return CQS_HIP_OK;
}

int32_t submit_locked(cqs_hip_embedder* e, uint32_t B, const std::function<int32_t(cqs_hip_embedder::Slot&)>& pack,
                      uint64_t* ticket) {
    if (e->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    if (!e->finalized) return efail(e, CQS_HIP_ERR_INVALID, "embed: weights not finalized");
    if (!ticket) return efail(e, CQS_HIP_ERR_INVALID, "embed: null ticket");
    *ticket = 0;
    if (B == 0) return efail(e, CQS_HIP_ERR_INVALID, "embed: empty batch");
    cqs_hip_embedder::Slot* sl = nullptr;
    for (cqs_hip_embedder::Slot& c : e->slot)
        if (c.ticket == 0) { sl = &c; break; }
    if (!sl) return efail(e, CQS_HIP_ERR_INVALID, "embed: every submission slot is in flight (collect a ticket first)");
    E_TRY(e, hipSetDevice(e->device));
    int32_t rc = pack(*sl);
    if (rc != CQS_HIP_OK) return rc;
    sl->batch_chain = false;
    sl->rerun = false;
    // The context with fewer tickets in flight; ties alternate - so pipelined callers interleave two kernel chains on the
    // device as before, while ONE blocking call at a time always lands on context 0: the same 0.3 GB of activation scratch
    // (and the same captured query graphs) call after call instead of two sets taking turns in L2 / the Infinity Cache.
    int load[cqs_hip_embedder::kCtx] = {0, 0};
    for (const cqs_hip_embedder::Slot& s2 : e->slot)
        if (s2.ticket != 0) load[s2.ctx]++;
    int ci = load[0] == load[1] ? (load[0] == 0 ? 0 : 1 - e->last_ctx) : (load[0] < load[1] ? 0 : 1);
    // A batch whose kernels are EXACT rounds of the 256 CUs (16 384 or 32 768 tokens: 128 / 256 row blocks x 2 workgroups) leaves a
    // second chain nothing to fill - side by side the two only contend.  Tickets in flight, chunks/s, two contexts | one | blocking
    // calls (a round-4 sweep script, since deleted, same box): 12 288 tokens 6 370 | 5 480 | 5 510; 14 336: 6 530 | 6 175 | 6 130; 16 384: 6 220-6 450 |
    // 6 400 | 6 440; 18 432: 5 975-6 300 | 4 845 | 4 820; 24 576: 6 050 | 5 690 | 5 740; 32 768: 6 340 | 6 530 | 6 600; 49 152: 6 220 | 6 110 |
    // 6 200; 65 536: 6 100 | 5 970 | 6 060.  So: those two sizes stay on context 0, everything else alternates.
    // The rule is the geometry, not the two bench shapes: the fused projections launch two workgroups per 128 token
    // rows; a batch is "exact rounds" when the last round of those workgroups is full or within 1/32 of full (16 384 - 500
    // tokens still is; 14 336 and 18 432 - 7/8 and 9/8 of a round - are not, and measured faster on two contexts above).
    {
        static const uint32_t n_cu = [] {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
            return (uint32_t)v;
        }();
        const uint32_t wgs = (sl->M + 127u) / 128u * 2u, last = wgs % n_cu;
        if (wgs >= n_cu && (last == 0u || last * 32u >= n_cu * 31u)) ci = 0;
    }
    if (e->single_ctx) ci = 0;            // CQS_HIP_EMBED_CONTEXTS=1: every ticket on one stream (tickets still overlap host packing / copies)
    e->last_ctx = ci;
    sl->ctx = ci;
    Ctx& c = e->ctx[ci];
    hipStream_t st = c.stream;
    const uint32_t H = e->g.hidden;
    if (sl->M == 0) {
        // every row empty: zero vectors (src/embedder/pooling.rs:113-119); still a ticket, nothing enqueued
        memset(sl->out, 0, (size_t)B * H * sizeof(float));
        E_TRY(e, hipEventRecord(sl->ev0, st));
        E_TRY(e, hipEventRecord(sl->ev1, st));
        E_TRY(e, hipEventRecord(sl->done, st));
    } else if (slot_takes_query_path(e, *sl)) {
        sl->direct = e->query_direct && load[ci] == 0;          // nothing else in flight on this context
        rc = run_query(e, c, *sl, sl->direct);
        if (rc != CQS_HIP_OK) { sl->direct = false; return rc; }
        E_TRY(e, hipEventRecord(sl->ev1, st));
        if (!sl->direct) E_TRY(e, hipMemcpyAsync(sl->out, c.q_out, (size_t)H * sizeof(float), hipMemcpyDeviceToHost, st));
        E_TRY(e, hipEventRecord(sl->done, st));
    } else {
        rc = enqueue_batch_chain(e, c, *sl);
        if (rc != CQS_HIP_OK) return rc;
        sl->batch_chain = true;
    }
    sl->ticket = e->next_ticket++;
    *ticket = sl->ticket;
    return CQS_HIP_OK;
}

}  // namespace

int32_t cqs_hip_embed_submit(cqs_hip_embedder* e, const int64_t* ids, const int64_t* mask, uint32_t B, uint32_t L,
                             uint64_t* ticket) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_embed_submit");
    if (!e) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!ids || !mask || L == 0) return efail(e, CQS_HIP_ERR_INVALID, "embed: null buffer / empty sequence");
    return submit_locked(e, B, [&](cqs_hip_embedder::Slot& sl) { return pack_padded(e, sl, ids, mask, B, L); }, ticket);
} CQS_ABI_CATCH(e)

int32_t cqs_hip_embed_submit_ragged(cqs_hip_embedder* e, const int32_t* tokens, const uint32_t* lens, uint32_t B,
                                    uint64_t* ticket) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_embed_submit_ragged");
    if (!e) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!lens || (!tokens && B)) return efail(e, CQS_HIP_ERR_INVALID, "embed: null buffer");
    return submit_locked(e, B, [&](cqs_hip_embedder::Slot& sl) { return pack_ragged(e, sl, tokens, lens, B); }, ticket);
} CQS_ABI_CATCH(e)

int32_t cqs_hip_embed_collect(cqs_hip_embedder* e, uint64_t ticket, float* out) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_embed_collect");
    if (!e) return CQS_HIP_ERR_INVALID;
    cqs_hip_embedder::Slot* sl = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (ticket == 0) return efail(e, CQS_HIP_ERR_INVALID, "collect: null ticket");
        for (cqs_hip_embedder::Slot& c : e->slot)
            if (c.ticket == ticket) { sl = &c; break; }
        if (!sl) return efail(e, CQS_HIP_ERR_INVALID, "collect: unknown ticket");
        if (sl->collecting) return efail(e, CQS_HIP_ERR_INVALID, "collect: this ticket is already being collected");
        sl->collecting = true;
    }
    // wait outside the lock: other threads may submit meanwhile; the slot stays ours until its ticket is cleared
    (void)hipSetDevice(e->device);
    const hipError_t he = hipEventSynchronize(sl->done);
    std::lock_guard<std::mutex> lk(e->mu);
    sl->collecting = false;
    if (he != hipSuccess) { sl->ticket = 0; return efail(e, CQS_HIP_ERR_DEVICE, "collect: device failure", he); }
    if (e->fuse_err && *(volatile unsigned*)e->fuse_err != 0u) {
        // A pair of the fused projection kernel never met (gemm_rowfuse.hip: workgroup b waits for granules of workgroup
        // b ^ 8 of the same launch, which needs both resident - true on an otherwise idle device, not promised by HIP):
        // rows of EVERY batch-chain ticket in flight may be garbage.  Round 5: the engine drops to the
        // two-launch chain (no cross-workgroup exchange, the same bits) for good and recomputes those tickets - this
        // one here, the others at their own collect - instead of poisoning itself.
        (void)hipDeviceSynchronize();
        *(volatile unsigned*)e->fuse_err = 0u;
        e->fuse_norm = 0;
        e->fuse_fallbacks++;
        for (cqs_hip_embedder::Slot& c2 : e->slot)
            if (c2.ticket != 0 && c2.batch_chain) c2.rerun = true;
    }
    if (sl->rerun) {
        sl->rerun = false;
        int32_t rc = enqueue_batch_chain(e, e->ctx[sl->ctx], *sl);
        if (rc == CQS_HIP_OK && hipEventSynchronize(sl->done) != hipSuccess) rc = efail(e, CQS_HIP_ERR_DEVICE, "collect: device failure in the recomputed batch");
        if (rc != CQS_HIP_OK) { sl->ticket = 0; return rc; }
    }
    const uint32_t H = e->g.hidden, B = sl->B;
    if (out) {                         // out == NULL: abandon the ticket (wait, release the slot, drop the rows)
        memcpy(out, sl->direct ? e->ctx[sl->ctx].q_out_pin : sl->out, (size_t)B * H * sizeof(float));
        const int32_t* seq_len = sl->meta + (size_t)2 * sl->M + B;
        for (uint32_t b = 0; b < B; ++b)   // empty rows: exact zeros, like the reference's zero-mask pooling
            if (seq_len[b] == 0) memset(out + (size_t)b * H, 0, (size_t)H * sizeof(float));
    }
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, sl->ev0, sl->ev1) == hipSuccess) e->last_ms = ms;
    sl->ticket = 0;
    sl->direct = false;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

// `session.run` (src/embedder/core.rs:1097): submit + collect.
int32_t cqs_hip_embed(cqs_hip_embedder* e, const int64_t* ids, const int64_t* mask, uint32_t B, uint32_t L, float* out) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_embed");
    if (!e) return CQS_HIP_ERR_INVALID;
    if (B == 0) return CQS_HIP_OK;
    if (!out) { std::lock_guard<std::mutex> lk(e->mu); return efail(e, CQS_HIP_ERR_INVALID, "embed: null buffer / empty sequence"); }
    uint64_t t = 0;
    const int32_t rc = cqs_hip_embed_submit(e, ids, mask, B, L, &t);
    if (rc != CQS_HIP_OK) return rc;
    return cqs_hip_embed_collect(e, t, out);
} CQS_ABI_CATCH(e)

// `Embedder::warm()` (src/embedder/core.rs:933-957: pay first-call cost before the first real query).  The search-time
// chain keeps one captured hipGraph per (query length, execution context, variant); without this call each is built
// the first time its length is seen (an eager chain + capture + instantiate + first replay on that query's clock).
// Runs every length 1..max_tokens once eagerly (sets the kernels' launch attributes), then captures, instantiates and
// replays each graph once, on both contexts, with token id 0.  Needs a free submission slot; nothing may be in flight.
int32_t cqs_hip_embedder_warm(cqs_hip_embedder* e, uint32_t max_tokens) CQS_ABI_TRY {
    if (!e) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    if (!e->finalized) return efail(e, CQS_HIP_ERR_INVALID, "warm: weights not finalized");
    if (!e->query_path || max_tokens == 0) return CQS_HIP_OK;
    for (const cqs_hip_embedder::Slot& s2 : e->slot)
        if (s2.ticket != 0) return efail(e, CQS_HIP_ERR_INVALID, "warm: tickets are in flight (collect them first)");
    cqs_hip_embedder::Slot& sl = e->slot[0];
    E_TRY(e, hipSetDevice(e->device));
    const uint32_t Tmax = std::min(max_tokens, std::min(e->query_max_tokens, e->g.max_seq));
    int32_t rc = slot_reserve(e, sl, 1, Tmax, 1);
    if (rc != CQS_HIP_OK) return rc;
    memset(sl.meta, 0, sl.meta_cap * sizeof(int32_t));          // token id 0 at every position
    const int n_ctx = e->single_ctx ? 1 : cqs_hip_embedder::kCtx;
    for (uint32_t T = 1; T <= Tmax; ++T) {
        sl.B = 1; sl.M = T; sl.nblk = 1;
        for (int ci = 0; ci < n_ctx; ++ci) {
            Ctx& c = e->ctx[ci];
            for (int var = 1; var >= 0; --var) {
                if (var == 1 && !e->query_direct) continue;
                for (int tries = 0; tries < 3; ++tries) {
                    if ((rc = run_query(e, c, sl, var == 1)) != CQS_HIP_OK) return rc;
                    if (c.q_exec[var][T - 1] || c.q_capture_failed[var][T - 1] || !e->query_graph) break;
                }
            }
            E_TRY(e, hipStreamSynchronize(c.stream));
        }
    }
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

// Search-time chain counters since the engine was made: graphs captured, captures that failed (those lengths run
// eagerly; the cause is in last_error), graph replays, eager chain runs.  Any pointer may be NULL.
void cqs_hip_embedder_query_graph_stats(const cqs_hip_embedder* e, uint64_t* captured, uint64_t* failed, uint64_t* replays,
                                        uint64_t* eager) CQS_ABI_TRY {
    uint64_t v[4] = {0, 0, 0, 0};
    if (e) {
        std::lock_guard<std::mutex> lk(e->mu);
        v[0] = e->q_captured; v[1] = e->q_capture_failures; v[2] = e->q_replays; v[3] = e->q_eager;
    }
    if (captured) *captured = v[0];
    if (failed) *failed = v[1];
    if (replays) *replays = v[2];
    if (eager) *eager = v[3];
} CQS_ABI_CATCH_VOID

// Diagnostic twin (synchronous): final-norm hidden states [B, L, hidden], zeros at padded positions.
int32_t cqs_hip_embed_hidden(cqs_hip_embedder* e, const int64_t* ids, const int64_t* mask, uint32_t B, uint32_t L, float* out) CQS_ABI_TRY {
    if (!e) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    if (!e->finalized) return efail(e, CQS_HIP_ERR_INVALID, "embed: weights not finalized");
    if (B == 0) return CQS_HIP_OK;
    if (!ids || !mask || !out || L == 0) return efail(e, CQS_HIP_ERR_INVALID, "embed: null buffer / empty sequence");
    cqs_hip_embedder::Slot* sl = nullptr;
    for (cqs_hip_embedder::Slot& c : e->slot)
        if (c.ticket == 0) { sl = &c; break; }
    if (!sl) return efail(e, CQS_HIP_ERR_INVALID, "embed: every submission slot is in flight (collect a ticket first)");
    E_TRY(e, hipSetDevice(e->device));
    int32_t rc = pack_padded(e, *sl, ids, mask, B, L);
    if (rc != CQS_HIP_OK) return rc;
    const uint32_t H = e->g.hidden;
    memset(out, 0, (size_t)B * L * H * sizeof(float));
    if (sl->M == 0) return CQS_HIP_OK;
    Ctx& c = e->ctx[0];
    rc = run_layers(e, c, *sl);
    if (rc != CQS_HIP_OK) return rc;
    hipStream_t st = c.stream;
    E_TRY(e, hipEventRecord(sl->ev1, st));
    std::vector<float> packed((size_t)sl->M * H);
    E_TRY(e, hipMemcpyAsync(packed.data(), c.hidden, packed.size() * 4, hipMemcpyDeviceToHost, st));
    E_TRY(e, hipStreamSynchronize(st));
    const int32_t* seq_start = sl->meta + (size_t)2 * sl->M;
    const int32_t* seq_len = seq_start + B;
    for (uint32_t b = 0; b < B; ++b)
        memcpy(out + (size_t)b * L * H, packed.data() + (size_t)seq_start[b] * H, (size_t)seq_len[b] * H * 4);
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, sl->ev0, sl->ev1) == hipSuccess) e->last_ms = ms;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

// `normalize_l2` (src/embedder/pooling.rs:60-67) over the rows of a host matrix: norm_sq folded left to right in
// f32, scaled by 1/sqrt when > 0, zero rows stay zero.  Host helper for callers that keep embeddings in one array.
void cqs_hip_normalize_l2_rows(float* rows, uint64_t n, uint32_t dim) CQS_ABI_TRY {
#pragma clang fp contract(off)   // Rust does not fuse x * x + acc: keep the reference's roundings
    if (!rows) return;
    for (uint64_t r = 0; r < n; ++r) {
        float* v = rows + r * dim;
        float norm_sq = 0.f;
        for (uint32_t i = 0; i < dim; ++i) norm_sq += v[i] * v[i];
        if (norm_sq > 0.f) {
            const float inv = 1.0f / sqrtf(norm_sq);
            for (uint32_t i = 0; i < dim; ++i) v[i] *= inv;
        }
    }
} CQS_ABI_CATCH_VOID

// Test hook (not part of the public header): the fused projection + add + norms kernel on / off and the token count it
// starts at, for THIS engine (the environment is read once, at finalize).  min_rows = 0 keeps the current threshold.
void cqs_hip_debug_embedder_set_fuse_norm(cqs_hip_embedder* e, int32_t on, uint32_t min_rows) CQS_ABI_TRY {
    if (!e) return;
    std::lock_guard<std::mutex> lk(e->mu);
    e->fuse_norm = on < 0 ? 0 : (on > 2 ? 2 : on);     // 0 = two launches, 1 = the 64-row kernel, 2 = the pair-split kernel
    if (min_rows) e->fuse_min_rows = min_rows;
} CQS_ABI_CATCH_VOID

// Test hook (not part of the public header): pretend a pair exchange of the fused projection kernel timed out (what the
// kernel's bounded spin reports through the pinned error word), and read how often the engine has fallen back.
void cqs_hip_debug_embedder_fake_fuse_timeout(cqs_hip_embedder* e) CQS_ABI_TRY {
    if (!e || !e->fuse_err) return;
    std::lock_guard<std::mutex> lk(e->mu);
    *(volatile unsigned*)e->fuse_err = 1u;
} CQS_ABI_CATCH_VOID
uint32_t cqs_hip_debug_embedder_fuse_fallbacks(cqs_hip_embedder* e) CQS_ABI_TRY {
    if (!e) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    return e->fuse_fallbacks;
} CQS_ABI_CATCH_VAL(0)

// Test hook (not part of the public header): the 256-row kernel's in-register GeGLU pairing on / off for THIS engine.
void cqs_hip_debug_embedder_set_geglu4(cqs_hip_embedder* e, int32_t on) CQS_ABI_TRY {
    if (!e) return;
    std::lock_guard<std::mutex> lk(e->mu);
    e->geglu4 = on != 0;
} CQS_ABI_CATCH_VOID

// Test hook (not part of the public header): the QKV projection's fused norm + RoPE epilogue on / off for THIS engine.
void cqs_hip_debug_embedder_set_fuse_qkv(cqs_hip_embedder* e, int32_t on) CQS_ABI_TRY {
    if (!e) return;
    std::lock_guard<std::mutex> lk(e->mu);
    e->fuse_qkv = on != 0;
} CQS_ABI_CATCH_VOID

// Diagnostic (not part of the public header; CQS_HIP_QUERY_STAMPS=1 at engine creation): the stamps the query chain's
// workgroups left in context `ctx` ([kernel slot][256][8] u64).  Returns the number of u64 copied.
uint64_t cqs_hip_debug_query_stamps(cqs_hip_embedder* e, uint32_t ctx, unsigned long long* out, uint64_t cap) CQS_ABI_TRY {
    if (!e || !out || ctx >= (uint32_t)cqs_hip_embedder::kCtx) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    Ctx& c = e->ctx[ctx];
    if (!c.q_dbg) return 0;
    const uint64_t words = std::min<uint64_t>(cap, ((uint64_t)e->g.layers * 5 + 2) * 256 * 8 * 2);
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(c.stream) != hipSuccess ||
        hipMemcpy(out, c.q_dbg, words * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return words;
} CQS_ABI_CATCH_VAL(0)

// Test / tuning aid (not part of the public header): one GEMM launch on caller-provided device buffers
// (bf16 A [M,K], bf16 W [N,K], C per out_kind) on `stream`; the kernel is chosen like in the forward
// (CQS_HIP_GEMM_TILE forces one).
int32_t cqs_hip_debug_gemm_run(const void* A, const void* W, void* C, uint32_t M, uint32_t N, uint32_t K, uint32_t ldc,
                               int32_t out_kind, void* stream) CQS_ABI_TRY {
    const hipError_t e = (out_kind & 0x100)   // + 0x100: the Dense head's skinny kernel
                             ? cqs::launch_gemm_skinny((const bf16_t*)A, (const bf16_t*)W, C, M, N, K, ldc,
                                                       (cqs::GemmOut)(out_kind & 0xff), (hipStream_t)stream)
                             : cqs::launch_gemm_bf16((const bf16_t*)A, (const bf16_t*)W, C, M, N, K, ldc, (cqs::GemmOut)out_kind,
                                                     (hipStream_t)stream);
    return e == hipSuccess ? CQS_HIP_OK : CQS_HIP_ERR_DEVICE;
} CQS_ABI_CATCH_NOHANDLE

// Tuning aid (not part of the public header): average milliseconds of one C[M,N] = A[M,K] W[N,K]^T
// launch over `iters` back-to-back launches on random bf16 operands.
float cqs_hip_debug_gemm_ms(uint32_t M, uint32_t N, uint32_t K, uint32_t iters, int32_t out_kind) CQS_ABI_TRY {
    bf16_t *A = nullptr, *W = nullptr;
    void* C = nullptr;
    if (dmalloc(&A, (size_t)M * K) != hipSuccess || dmalloc(&W, (size_t)N * K) != hipSuccess ||
        hipMalloc(&C, (size_t)M * N * 4) != hipSuccess) return -1.f;
    std::vector<uint16_t> h((size_t)std::max(M, N) * K);
    uint32_t st = 12345u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = f32_to_bf16_bits(((st >> 8) & 0xFFFF) / 32768.0f - 1.0f); }
    (void)hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const cqs::GemmOut ok = (cqs::GemmOut)out_kind;
    const uint32_t ldc = ok == cqs::GEMM_OUT_GEGLU ? N / 2 : N;
    (void)cqs::launch_gemm_bf16(A, W, C, M, N, K, ldc, ok, nullptr);
    (void)hipEventRecord(e0, nullptr);
    for (uint32_t i = 0; i < iters; ++i) (void)cqs::launch_gemm_bf16(A, W, C, M, N, K, ldc, ok, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / (float)iters;
} CQS_ABI_CATCH_VAL(-1.f)

}  // extern "C"
