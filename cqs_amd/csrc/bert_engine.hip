// bert_engine.hip — the BERT-family engine of libcqs_hip.so (include/cqs_hip.h, "BERT-family auxiliary models"):
// the device side of cqs's SPLADE sparse encoder (`SpladeEncoder::encode_batch`, src/splade/mod.rs:774-1075: ORT runs a
// BERT masked-LM, Rust pools its logits) and of its cross-encoder reranker (`compute_scores_opt`, src/reranker.rs:343-
// 533: ORT runs BertForSequenceClassification, Rust applies a sigmoid).  SURVEY.md §8(f)4.  Operator semantics:
// oracle/bert_ref.py.  No CPU fallback: without a GPU `cqs_hip_bert_create` fails.
#include "../../include/cqs_hip.h"
#include "abi_guard.h"
#include "roctx.h"
#include "bert_kernels.h"
#include "onnx_reader.h"
#include "safetensors_reader.h"

#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

using cqs::bf16_t;

namespace {

uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

struct BertLayer {
    bf16_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
};

}  // namespace

struct cqs_hip_bert {
    int device = 0;
    cqs_hip_bert_config cfg{};
    uint32_t vpad = 0;                       // vocab rounded up to a multiple of 192 (decoder N tile)
    std::map<std::string, std::vector<float>> pending;    // host copies until finalize
    bool finalized = false;

    bf16_t *word = nullptr, *posw = nullptr, *typew = nullptr;     // word: [vpad, H] (rows >= vocab are zero)
    float *emb_g = nullptr, *emb_b = nullptr;
    std::vector<BertLayer> L;
    // masked-LM head
    bf16_t* wt = nullptr;
    float *bt = nullptr, *lnt_g = nullptr, *lnt_b = nullptr, *bdec = nullptr;   // bdec [vpad]
    // classifier head
    bf16_t *wp = nullptr, *wc = nullptr;     // wc [16-padded labels, H]
    float *bp = nullptr, *bc = nullptr;
    std::vector<void*> owned;                // every device allocation of the weights

    // Execution contexts (a HIP stream + the activation scratch of one batch each; weights shared): consecutive tickets
    // alternate, so batch i + 1's kernel chain fills the CUs batch i's leaves idle and its H2D / host packing overlap
    // batch i's compute - the submit / collect scheme of the EmbeddingGemma engine (embedder.hip), which the reference's
    // index pipeline needs from the SPLADE encoder just as well (src/splade/mod.rs:774-1075 is called per batch from it).
    struct Ctx {
        hipStream_t stream = nullptr;
        uint32_t tok_cap = 0, seq_cap = 0, blk_cap = 0;
        bf16_t *x = nullptr, *x2 = nullptr, *y = nullptr, *qkv = nullptr, *att = nullptr, *h = nullptr, *pooled = nullptr;   // x2: the other residual buffer of the fused small-batch chain
        float *dense = nullptr, *cls = nullptr;
        uint32_t *sp_ids = nullptr, *sp_cnt = nullptr;  // device-side threshold filter: [sp_rows * sp_cap] ids / weights, [sp_rows] counts
        float* sp_w = nullptr;
        size_t sp_rows = 0, sp_cap = 0;
        int32_t* d_meta = nullptr;
    };
    static constexpr int kCtx = 2;
    Ctx ctx[kCtx];
    // Submission slots: pinned tables (packed on the host while earlier tickets run) + pinned results + a completion event.
    struct Slot {
        uint64_t ticket = 0;         // 0 = free
        int kind = 0;                // 1 = sparse SPLADE vectors, 2 = pooled embeddings
        int ctx = 0;
        uint32_t B = 0, M = 0, cap = 0, nblk = 0;
        int32_t* meta = nullptr;     // pinned: [tok M][pos M][tt M][seq_start B][seq_len B][blk 2 nblk][row_seq M]
        size_t meta_cap = 0;         // int32 elements
        void* out = nullptr;         // pinned results: kind 1: ids [B cap] u32 | weights [B cap] f32 | counts [B] u32; kind 2: [B, H] f32
        size_t out_cap = 0;          // bytes
        hipEvent_t done = nullptr;
        bool collecting = false;     // a collect is waiting on this ticket (a second collector of the same ticket is refused)
    };
    // Three TICKET slots (the public contract: up to 3 tickets in flight) + one slot RESERVED for the blocking calls
    // (cqs_hip_splade_encode, _encode_sparse, cqs_hip_rerank_logits, cqs_hip_bert_embed, _hidden), so that a search-time
    // query or rerank never fails because an index run holds every ticket; a blocking call that finds the reserved slot
    // taken by another thread's blocking call waits for it (cv).
    static constexpr int kSlots = 3;
    static constexpr int kReserved = kSlots;
    Slot slot[kSlots + 1];
    std::condition_variable slot_cv;
    uint64_t next_ticket = 1;
    int last_ctx = 1;                    // context of the previous ticket (ties alternate)

    std::mutex mu;
    std::atomic<bool> poisoned{false};
    std::string last_error;
};

namespace {

int32_t bfail(cqs_hip_bert* e, int32_t code, const std::string& what, hipError_t he = hipSuccess) {
    std::string msg = what;
    if (he != hipSuccess) msg += std::string(": ") + hipGetErrorString(he);
    if (e) {
        e->last_error = msg;
        if (code == CQS_HIP_ERR_DEVICE) e->poisoned.store(true, std::memory_order_release);
    }
    return code;
}
#define B_TRY(e, expr)                                                                                                   \
    do {                                                                                                                 \
        hipError_t _h = (expr);                                                                                          \
        if (_h != hipSuccess) return bfail((e), _h == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE, #expr, _h); \
    } while (0)

const std::vector<float>* find(cqs_hip_bert* e, const std::string& name, size_t count) {
    auto it = e->pending.find(name);
    if (it == e->pending.end() || it->second.size() != count) return nullptr;
    return &it->second;
}

int32_t up_bf16(cqs_hip_bert* e, bf16_t** dst, const float* src, size_t count, size_t alloc_count = 0) {
    if (alloc_count < count) alloc_count = count;
    std::vector<uint16_t> tmp(alloc_count, 0);
    for (size_t i = 0; i < count; ++i) tmp[i] = f32_to_bf16(src[i]);
    B_TRY(e, hipMalloc((void**)dst, alloc_count * 2));
    e->owned.push_back(*dst);
    B_TRY(e, hipMemcpy(*dst, tmp.data(), alloc_count * 2, hipMemcpyHostToDevice));
    return CQS_HIP_OK;
}
int32_t up_f32(cqs_hip_bert* e, float** dst, const float* src, size_t count, size_t alloc_count = 0) {
    if (alloc_count < count) alloc_count = count;
    std::vector<float> tmp(alloc_count, 0.f);
    memcpy(tmp.data(), src, count * 4);
    B_TRY(e, hipMalloc((void**)dst, alloc_count * 4));
    e->owned.push_back(*dst);
    B_TRY(e, hipMemcpy(*dst, tmp.data(), alloc_count * 4, hipMemcpyHostToDevice));
    return CQS_HIP_OK;
}

using BCtx = cqs_hip_bert::Ctx;
using BSlot = cqs_hip_bert::Slot;

void free_scratch(BCtx& c) {
    void** all[] = {(void**)&c.x, (void**)&c.x2, (void**)&c.y, (void**)&c.qkv, (void**)&c.att, (void**)&c.h,
                    (void**)&c.pooled, (void**)&c.dense, (void**)&c.cls, (void**)&c.d_meta};
    for (void** p : all) { (void)hipFree(*p); *p = nullptr; }
    (void)hipFree(c.sp_ids); (void)hipFree(c.sp_w); (void)hipFree(c.sp_cnt);
    c.sp_ids = c.sp_cnt = nullptr; c.sp_w = nullptr; c.sp_rows = c.sp_cap = 0;
    c.tok_cap = c.seq_cap = c.blk_cap = 0;
}

int32_t ensure_scratch(cqs_hip_bert* e, BCtx& c, uint32_t M, uint32_t B, uint32_t nblk) {
    if (M <= c.tok_cap && B <= c.seq_cap && nblk <= c.blk_cap) return CQS_HIP_OK;
    B_TRY(e, hipStreamSynchronize(c.stream));
    const uint32_t Mc = std::max(M, c.tok_cap), Bc = std::max(B, c.seq_cap), bc = std::max(nblk, c.blk_cap);
    free_scratch(c);
    const cqs_hip_bert_config& cf = e->cfg;
    const size_t H = cf.hidden;
    B_TRY(e, hipMalloc((void**)&c.x, (size_t)Mc * H * 2));
    B_TRY(e, hipMalloc((void**)&c.x2, (size_t)std::min<uint32_t>(Mc, 64u) * H * 2));
    B_TRY(e, hipMalloc((void**)&c.y, (size_t)Mc * H * 2));
    B_TRY(e, hipMalloc((void**)&c.qkv, (size_t)Mc * 3 * H * 2));
    B_TRY(e, hipMalloc((void**)&c.att, (size_t)Mc * H * 2));
    B_TRY(e, hipMalloc((void**)&c.h, (size_t)Mc * cf.intermediate * 2));
    if (cf.head == CQS_HIP_BERT_HEAD_MLM) {
        B_TRY(e, hipMalloc((void**)&c.dense, (size_t)Bc * cf.vocab_size * 4));
    } else if (cf.head == CQS_HIP_BERT_HEAD_NONE) {
        B_TRY(e, hipMalloc((void**)&c.dense, (size_t)Bc * H * 4));            // pooled embeddings
    } else {
        B_TRY(e, hipMalloc((void**)&c.pooled, (size_t)Bc * H * 2));
        B_TRY(e, hipMalloc((void**)&c.cls, (size_t)Bc * 16 * 4));
    }
    B_TRY(e, hipMalloc((void**)&c.d_meta, ((size_t)4 * Mc + (size_t)2 * Bc + (size_t)2 * bc) * 4));
    c.tok_cap = Mc; c.seq_cap = Bc; c.blk_cap = bc;
    return CQS_HIP_OK;
}

// Validate + pack a ragged batch into the slot's pinned tables, upload them, run the encoder on context `c`; leaves the
// final hidden states in c.x.
int32_t run_encoder(cqs_hip_bert* e, BCtx& c, BSlot& sl, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens, uint32_t B,
                    uint32_t* M_out) {
    const cqs_hip_bert_config& cf = e->cfg;
    uint64_t M64 = 0, nblk64 = 0;
    uint32_t max_len = 0;
    for (uint32_t b = 0; b < B; ++b) {
        if (lens[b] > cf.max_pos) return bfail(e, CQS_HIP_ERR_INVALID, "bert: sequence longer than max_position_embeddings");
        M64 += lens[b];
        nblk64 += (lens[b] + 63u) / 64u;
        max_len = std::max(max_len, lens[b]);
    }
    if (M64 > 0x7FFFFFFFull) return bfail(e, CQS_HIP_ERR_INVALID, "bert: batch holds too many tokens");
    const uint32_t M = (uint32_t)M64, nblk = (uint32_t)nblk64;
    *M_out = M;
    sl.B = B; sl.M = M; sl.nblk = nblk;
    if (M == 0) return CQS_HIP_OK;
    // tables: [tok M][pos M][tt M][seq_start B][seq_len B][blk 2 nblk][row_seq M]
    const size_t words = (size_t)4 * M + (size_t)2 * B + (size_t)2 * nblk;
    if (words > sl.meta_cap) {
        if (sl.meta) (void)hipHostFree(sl.meta);
        sl.meta = nullptr; sl.meta_cap = 0;
        const size_t cap = words + words / 4 + 64;
        B_TRY(e, hipHostMalloc((void**)&sl.meta, cap * sizeof(int32_t), hipHostMallocDefault));
        sl.meta_cap = cap;
    }
    int32_t *tok = sl.meta, *pos = tok + M, *tt = pos + M, *seq_start = tt + M, *seq_len = seq_start + B, *blk = seq_len + B;
    int32_t* row_seq = blk + (size_t)2 * nblk;
    uint32_t m = 0, nb = 0;
    for (uint32_t b = 0; b < B; ++b) {
        seq_start[b] = (int32_t)m;
        seq_len[b] = (int32_t)lens[b];
        for (uint32_t j = 0; j < lens[b]; ++j) {
            const int32_t id = tokens[m + j];
            if (id < 0 || (uint32_t)id >= cf.vocab_size) return bfail(e, CQS_HIP_ERR_INVALID, "bert: token id out of range");
            const int32_t ty = type_ids ? type_ids[m + j] : 0;
            if (ty < 0 || (uint32_t)ty >= cf.type_vocab) return bfail(e, CQS_HIP_ERR_INVALID, "bert: token type id out of range");
            tok[m + j] = id; pos[m + j] = (int32_t)j; tt[m + j] = ty; row_seq[m + j] = (int32_t)b;
        }
        for (uint32_t q = 0; q * 64u < lens[b]; ++q) { blk[2 * nb] = (int32_t)b; blk[2 * nb + 1] = (int32_t)q; ++nb; }
        m += lens[b];
    }
    int32_t rc = ensure_scratch(e, c, M, B, nblk);
    if (rc != CQS_HIP_OK) return rc;
    hipStream_t st = c.stream;
    B_TRY(e, hipMemcpyAsync(c.d_meta, sl.meta, words * 4, hipMemcpyHostToDevice, st));
    const int32_t *d_tok = c.d_meta, *d_pos = d_tok + M, *d_tt = d_pos + M, *d_start = d_tt + M, *d_len = d_start + B,
                  *d_blk = d_len + B;
    const uint32_t H = cf.hidden, I = cf.intermediate;
    B_TRY(e, cqs::launch_bert_embed_ln(d_tok, d_pos, d_tt, e->word, e->posw, e->typew, e->emb_g, e->emb_b, cf.ln_eps, c.x, M, H, st));
    // Up to 64 tokens (a search-time query): 5 launches per layer instead of 7 - each residual add + LayerNorm runs in the
    // prologue of the projection that consumes it (launch_gemm_small_rows_addln: every workgroup normalises the rows it
    // multiplies; the stream alternates between two buffers because everybody reads the old one)
    {
        const char* sr = getenv("CQS_HIP_GEMM_SMALL_ROWS");       // (the switch of the small-rows kernels covers this chain too)
        const bool fused = M <= 64u && !(sr && sr[0] == '0') && (H == 768u || H == 1024u || H == 256u) && H % 16u == 0 && I % 16u == 0;
        if (fused) {
            bf16_t* cur = c.x;
            bf16_t* oth = c.x2;
            for (uint32_t l = 0; l < cf.layers; ++l) {
                const BertLayer& w = e->L[l];
                if (l == 0) {
                    B_TRY(e, cqs::launch_gemm_bias(cur, w.wqkv, w.bqkv, c.qkv, M, 3u * H, H, 3u * H, cqs::GEMM_OUT_BF16, st));
                } else {
                    const BertLayer& pv = e->L[l - 1];
                    B_TRY(e, cqs::launch_gemm_small_rows_addln(cur, c.y, pv.ln2_g, pv.ln2_b, cf.ln_eps, oth, w.wqkv, w.bqkv, c.qkv, M, 3u * H, H,
                                                               3u * H, cqs::GEMM_OUT_BF16, st));
                    std::swap(cur, oth);
                }
                B_TRY(e, cqs::launch_bert_attention(c.qkv, c.att, d_blk, nblk, d_start, d_len, B, max_len, cf.heads, H / cf.heads, st));
                B_TRY(e, cqs::launch_gemm_bias(c.att, w.wo, w.bo, c.y, M, H, H, H, cqs::GEMM_OUT_BF16, st));
                B_TRY(e, cqs::launch_gemm_small_rows_addln(cur, c.y, w.ln1_g, w.ln1_b, cf.ln_eps, oth, w.w1, w.b1, c.h, M, I, H, I,
                                                           cqs::GEMM_OUT_BF16_GELU, st));
                std::swap(cur, oth);
                B_TRY(e, cqs::launch_gemm_bias(c.h, w.w2, w.b2, c.y, M, H, I, H, cqs::GEMM_OUT_BF16, st));
            }
            const BertLayer& lw = e->L[cf.layers - 1u];
            B_TRY(e, cqs::launch_bert_add_ln(cur, c.y, lw.ln2_g, lw.ln2_b, cf.ln_eps, c.x, M, H, st));     // the final hidden states, in c.x
            return CQS_HIP_OK;
        }
    }
    for (uint32_t l = 0; l < cf.layers; ++l) {
        const BertLayer& w = e->L[l];
        B_TRY(e, cqs::launch_gemm_bias(c.x, w.wqkv, w.bqkv, c.qkv, M, 3u * H, H, 3u * H, cqs::GEMM_OUT_BF16, st));
        B_TRY(e, cqs::launch_bert_attention(c.qkv, c.att, d_blk, nblk, d_start, d_len, B, max_len, cf.heads, H / cf.heads, st));
        B_TRY(e, cqs::launch_gemm_bias(c.att, w.wo, w.bo, c.y, M, H, H, H, cqs::GEMM_OUT_BF16, st));
        B_TRY(e, cqs::launch_bert_add_ln(c.x, c.y, w.ln1_g, w.ln1_b, cf.ln_eps, c.x, M, H, st));
        B_TRY(e, cqs::launch_gemm_bias(c.x, w.w1, w.b1, c.h, M, I, H, I, cqs::GEMM_OUT_BF16_GELU, st));
        B_TRY(e, cqs::launch_gemm_bias(c.h, w.w2, w.b2, c.y, M, H, I, H, cqs::GEMM_OUT_BF16, st));
        B_TRY(e, cqs::launch_bert_add_ln(c.x, c.y, w.ln2_g, w.ln2_b, cf.ln_eps, c.x, M, H, st));
    }
    return CQS_HIP_OK;
}

// Set (per thread) by the blocking forms around their submit: the submission takes the reserved slot.
thread_local bool tl_blocking_call = false;
struct BlockingScope {
    BlockingScope() { tl_blocking_call = true; }
    ~BlockingScope() { tl_blocking_call = false; }
};
// The reserved slot, waiting (mu released meanwhile) while another thread's blocking call holds it.
void wait_reserved(cqs_hip_bert* e, std::unique_lock<std::mutex>& lk) {
    e->slot_cv.wait(lk, [&] { return e->slot[cqs_hip_bert::kReserved].ticket == 0; });
}
// A free submission slot + the context the next ticket runs on (consecutive tickets alternate).  reserved: the blocking
// calls' own slot (the caller has waited for it under the lock).
int32_t take_slot(cqs_hip_bert* e, BSlot** sl, BCtx** c, bool reserved = false) {
    *sl = nullptr;
    if (reserved) *sl = &e->slot[cqs_hip_bert::kReserved];
    else
        for (int i = 0; i < cqs_hip_bert::kSlots; ++i)
            if (e->slot[i].ticket == 0) { *sl = &e->slot[i]; break; }
    if (!*sl) return bfail(e, CQS_HIP_ERR_INVALID, "bert: every submission slot is in flight (collect a ticket first)");
    // the context with fewer tickets in flight; ties alternate, an idle engine takes context 0 (one blocking call at a time
    // keeps re-using ONE set of activation scratch instead of two taking turns in L2 / the Infinity Cache)
    int load[cqs_hip_bert::kCtx] = {0, 0};
    for (const BSlot& s2 : e->slot)
        if (s2.ticket != 0) load[s2.ctx]++;
    const int ci = load[0] == load[1] ? (load[0] == 0 ? 0 : 1 - e->last_ctx) : (load[0] < load[1] ? 0 : 1);
    e->last_ctx = ci;
    (*sl)->ctx = ci;
    *c = &e->ctx[ci];
    if (!(*sl)->done) B_TRY(e, hipEventCreateWithFlags(&(*sl)->done, hipEventDisableTiming));
    return CQS_HIP_OK;
}
int32_t slot_out_reserve(cqs_hip_bert* e, BSlot& sl, size_t bytes) {
    if (bytes <= sl.out_cap) return CQS_HIP_OK;
    if (sl.out) (void)hipHostFree(sl.out);
    sl.out = nullptr; sl.out_cap = 0;
    const size_t cap = bytes + bytes / 4 + 256;
    B_TRY(e, hipHostMalloc(&sl.out, cap, hipHostMallocDefault));
    sl.out_cap = cap;
    return CQS_HIP_OK;
}
BSlot* find_ticket(cqs_hip_bert* e, uint64_t ticket, int kind) {
    if (ticket == 0) return nullptr;
    for (BSlot& s : e->slot)
        if (s.ticket == ticket && s.kind == kind) return &s;
    return nullptr;
}

int32_t check_ready(cqs_hip_bert* e, uint32_t head) {
    if (e->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    if (!e->finalized) return bfail(e, CQS_HIP_ERR_INVALID, "bert: weights not finalized");
    if (head != 0xFFFFFFFFu && e->cfg.head != head) return bfail(e, CQS_HIP_ERR_INVALID, "bert: this engine was built with the other head");
    return CQS_HIP_OK;
}

}  // namespace

extern "C" {

int32_t cqs_hip_bert_config_default(uint32_t head, cqs_hip_bert_config* c) CQS_ABI_TRY {
    if (!c) return CQS_HIP_ERR_INVALID;
    memset(c, 0, sizeof(*c));
    c->vocab_size = 30522; c->max_pos = 512; c->type_vocab = 2; c->ln_eps = 1e-12f; c->num_labels = 1; c->head = head;
    if (head == CQS_HIP_BERT_HEAD_MLM) {            // naver/splade-cocondenser-ensembledistil = bert-base-uncased geometry
        c->hidden = 768; c->layers = 12; c->heads = 12; c->intermediate = 3072;
    } else if (head == CQS_HIP_BERT_HEAD_CLASSIFIER) {   // cross-encoder/ms-marco-MiniLM-L-6-v2 (src/reranker.rs:7,35)
        c->hidden = 384; c->layers = 6; c->heads = 12; c->intermediate = 1536;
    } else if (head == CQS_HIP_BERT_HEAD_NONE) {         // intfloat/e5-base-v2 = BERT-base (src/embedder/models.rs:346-358)
        c->hidden = 768; c->layers = 12; c->heads = 12; c->intermediate = 3072;
    } else {
        return CQS_HIP_ERR_INVALID;
    }
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

int32_t cqs_hip_bert_create(const cqs_hip_bert_config* cfg, int32_t device, cqs_hip_bert** out) CQS_ABI_TRY {
    if (!cfg || !out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    const cqs_hip_bert_config& c = *cfg;
    auto tiles = [](uint32_t n) { return n && (n % 192u == 0 || n % 256u == 0 || n % 320u == 0); };   // a GEMM tile width divides it
    const bool ok_cfg = c.hidden && c.hidden % 128u == 0 && c.hidden <= 1024u && c.layers && c.heads &&
                        c.hidden % c.heads == 0 && (c.hidden / c.heads == 32u || c.hidden / c.heads == 64u) &&
                        tiles(c.hidden) && tiles(3u * c.hidden) && tiles(c.intermediate) && c.vocab_size && c.max_pos && c.type_vocab &&
                        (c.head == CQS_HIP_BERT_HEAD_MLM || c.head == CQS_HIP_BERT_HEAD_CLASSIFIER || c.head == CQS_HIP_BERT_HEAD_NONE) &&
                        (c.head != CQS_HIP_BERT_HEAD_CLASSIFIER || (c.num_labels >= 1 && c.num_labels <= 16)) && c.ln_eps > 0.f;
    if (!ok_cfg) return CQS_HIP_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return CQS_HIP_ERR_NO_DEVICE;
    cqs_hip_bert* e = new (std::nothrow) cqs_hip_bert();
    if (!e) return CQS_HIP_ERR_NOMEM;
    std::unique_ptr<cqs_hip_bert, void (*)(cqs_hip_bert*)> owner(e, cqs_hip_bert_destroy);   // every failing exit gives the handle (and its streams) back
    e->device = device;
    e->cfg = c;
    e->vpad = (c.vocab_size + 191u) / 192u * 192u;
    e->L.resize(c.layers);
    if (hipSetDevice(device) != hipSuccess) return CQS_HIP_ERR_DEVICE;
    for (cqs_hip_bert::Ctx& cx : e->ctx)
        if (hipStreamCreateWithFlags(&cx.stream, hipStreamNonBlocking) != hipSuccess) return CQS_HIP_ERR_DEVICE;
    *out = owner.release();
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

// HF names, with or without the leading `bert.`; data f32, row-major; copied.
int32_t cqs_hip_bert_set_tensor(cqs_hip_bert* e, const char* name, const float* data, uint64_t count) CQS_ABI_TRY {
    if (!e || !name || !data) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->finalized) return bfail(e, CQS_HIP_ERR_INVALID, "bert: weights already finalized");
    std::string n(name);
    if (n.rfind("bert.", 0) == 0) n = n.substr(5);
    e->pending[n].assign(data, data + count);
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

int32_t cqs_hip_bert_finalize(cqs_hip_bert* e) CQS_ABI_TRY {
    if (!e) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->finalized) return CQS_HIP_OK;
    B_TRY(e, hipSetDevice(e->device));
    const cqs_hip_bert_config& c = e->cfg;
    const size_t H = c.hidden, I = c.intermediate, V = c.vocab_size;
    auto need = [&](const std::string& name, size_t count) -> const float* {
        const std::vector<float>* v = find(e, name, count);
        if (!v) bfail(e, CQS_HIP_ERR_INVALID, "bert: missing or mis-sized tensor " + name);
        return v ? v->data() : nullptr;
    };
    int32_t rc;
#define NEED(var, name, count) const float* var = need(name, count); if (!var) return CQS_HIP_ERR_INVALID
#define UP(call) if ((rc = (call)) != CQS_HIP_OK) return rc
    NEED(word, "embeddings.word_embeddings.weight", V * H);
    NEED(posw, "embeddings.position_embeddings.weight", (size_t)c.max_pos * H);
    NEED(typew, "embeddings.token_type_embeddings.weight", (size_t)c.type_vocab * H);
    NEED(eg, "embeddings.LayerNorm.weight", H);
    NEED(eb, "embeddings.LayerNorm.bias", H);
    UP(up_bf16(e, &e->word, word, V * H, (size_t)e->vpad * H));      // zero rows up to vpad: the tied decoder's padded N
    UP(up_bf16(e, &e->posw, posw, (size_t)c.max_pos * H));
    UP(up_bf16(e, &e->typew, typew, (size_t)c.type_vocab * H));
    UP(up_f32(e, &e->emb_g, eg, H));
    UP(up_f32(e, &e->emb_b, eb, H));
    for (uint32_t l = 0; l < c.layers; ++l) {
        const std::string p = "encoder.layer." + std::to_string(l) + ".";
        BertLayer& w = e->L[l];
        NEED(wq, p + "attention.self.query.weight", H * H);
        NEED(wk, p + "attention.self.key.weight", H * H);
        NEED(wv, p + "attention.self.value.weight", H * H);
        NEED(bq, p + "attention.self.query.bias", H);
        NEED(bk, p + "attention.self.key.bias", H);
        NEED(bv, p + "attention.self.value.bias", H);
        std::vector<float> fw(3 * H * H), fb(3 * H);
        memcpy(fw.data(), wq, H * H * 4); memcpy(fw.data() + H * H, wk, H * H * 4); memcpy(fw.data() + 2 * H * H, wv, H * H * 4);
        memcpy(fb.data(), bq, H * 4); memcpy(fb.data() + H, bk, H * 4); memcpy(fb.data() + 2 * H, bv, H * 4);
        UP(up_bf16(e, &w.wqkv, fw.data(), fw.size()));
        UP(up_f32(e, &w.bqkv, fb.data(), fb.size()));
        NEED(wo, p + "attention.output.dense.weight", H * H);
        NEED(bo, p + "attention.output.dense.bias", H);
        NEED(g1, p + "attention.output.LayerNorm.weight", H);
        NEED(b1n, p + "attention.output.LayerNorm.bias", H);
        NEED(w1, p + "intermediate.dense.weight", I * H);
        NEED(b1, p + "intermediate.dense.bias", I);
        NEED(w2, p + "output.dense.weight", H * I);
        NEED(b2, p + "output.dense.bias", H);
        NEED(g2, p + "output.LayerNorm.weight", H);
        NEED(b2n, p + "output.LayerNorm.bias", H);
        UP(up_bf16(e, &w.wo, wo, H * H)); UP(up_f32(e, &w.bo, bo, H));
        UP(up_f32(e, &w.ln1_g, g1, H)); UP(up_f32(e, &w.ln1_b, b1n, H));
        UP(up_bf16(e, &w.w1, w1, I * H)); UP(up_f32(e, &w.b1, b1, I));
        UP(up_bf16(e, &w.w2, w2, H * I)); UP(up_f32(e, &w.b2, b2, H));
        UP(up_f32(e, &w.ln2_g, g2, H)); UP(up_f32(e, &w.ln2_b, b2n, H));
    }
    if (c.head == CQS_HIP_BERT_HEAD_MLM) {
        NEED(wt, "cls.predictions.transform.dense.weight", H * H);
        NEED(bt, "cls.predictions.transform.dense.bias", H);
        NEED(lg, "cls.predictions.transform.LayerNorm.weight", H);
        NEED(lb, "cls.predictions.transform.LayerNorm.bias", H);
        NEED(bd, "cls.predictions.bias", V);
        UP(up_bf16(e, &e->wt, wt, H * H)); UP(up_f32(e, &e->bt, bt, H));
        UP(up_f32(e, &e->lnt_g, lg, H)); UP(up_f32(e, &e->lnt_b, lb, H));
        UP(up_f32(e, &e->bdec, bd, V, e->vpad));
    } else if (c.head == CQS_HIP_BERT_HEAD_CLASSIFIER) {
        NEED(wp, "pooler.dense.weight", H * H);
        NEED(bp, "pooler.dense.bias", H);
        NEED(wc, "classifier.weight", (size_t)c.num_labels * H);
        NEED(bc, "classifier.bias", c.num_labels);
        UP(up_bf16(e, &e->wp, wp, H * H)); UP(up_f32(e, &e->bp, bp, H));
        UP(up_bf16(e, &e->wc, wc, (size_t)c.num_labels * H, (size_t)16 * H));    // zero rows up to 16 labels
        UP(up_f32(e, &e->bc, bc, c.num_labels, 16));
    }
#undef NEED
#undef UP
    e->pending.clear();
    e->finalized = true;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

namespace {

// Encoder + masked-LM head + pooling + activation: leaves the [batch, vocab] activations in c.dense (device).
// *M_out == 0: every sequence empty (nothing was launched; the activations are all ln(1 + 0) = 0).
int32_t splade_forward(cqs_hip_bert* e, BCtx& c, BSlot& sl, const int32_t* tokens, const uint32_t* lens, uint32_t batch, uint32_t* M_out) {
    uint64_t tot = 0;
    for (uint32_t b = 0; b < batch; ++b) tot += lens[b];
    if (tot && !tokens) return bfail(e, CQS_HIP_ERR_INVALID, "splade: null tokens");
    int32_t rc = run_encoder(e, c, sl, tokens, nullptr, lens, batch, M_out);
    if (rc != CQS_HIP_OK) return rc;
    const uint32_t M = *M_out;
    if (M == 0) return CQS_HIP_OK;
    const cqs_hip_bert_config& cf = e->cfg;
    const size_t V = cf.vocab_size;
    hipStream_t st = c.stream;
    const uint32_t H = cf.hidden;
    const int32_t* d_len = c.d_meta + (size_t)3 * M + batch;
    // BertLMPredictionHead: LayerNorm(gelu(x Wt^T + bt)) E^T + b, decoder tied to the (zero-padded) word embeddings.
    // The [tokens, vocab] logits are never stored: the decoder GEMM's epilogue keeps, per sequence and vocabulary
    // entry, the maximum of max(0, logit) (atomic max on the float bits), then one small pass takes ln(1 + .).
    const int32_t* d_rowseq = d_len + batch + (size_t)2 * sl.nblk;
    B_TRY(e, cqs::launch_gemm_bias(c.x, e->wt, e->bt, c.y, M, H, H, H, cqs::GEMM_OUT_BF16_GELU, st));
    B_TRY(e, cqs::launch_bert_add_ln(c.y, nullptr, e->lnt_g, e->lnt_b, cf.ln_eps, c.y, M, H, st));
    B_TRY(e, hipMemsetAsync(c.dense, 0, (size_t)batch * V * 4, st));
    B_TRY(e, cqs::launch_gemm_rowmax(c.y, e->word, e->bdec, (uint32_t*)c.dense, M, e->vpad, H, (uint32_t)V, d_rowseq, (uint32_t)V, st));
    B_TRY(e, cqs::launch_splade_activate(c.dense, (size_t)batch * V, st));
    return CQS_HIP_OK;
}

}  // namespace

// `SpladeEncoder::encode_batch` below the tokenizer: sequences back to back (i32 ids) + their lengths; out_dense
// [batch, vocab] f32 = ln(1 + max(0, max over the sequence's tokens of the MLM logits)) - the model's pre-pooled
// `sparse_vector` output form (src/splade/mod.rs:960-978); the caller keeps entries > threshold.
int32_t cqs_hip_splade_encode(cqs_hip_bert* e, const int32_t* tokens, const uint32_t* lens, uint32_t batch, float* out_dense) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_splade_encode");
    if (!e) return CQS_HIP_ERR_INVALID;
    std::unique_lock<std::mutex> lk(e->mu);
    int32_t rc = check_ready(e, CQS_HIP_BERT_HEAD_MLM);
    if (rc != CQS_HIP_OK) return rc;
    if (batch == 0) return CQS_HIP_OK;
    if (!lens || !out_dense) return bfail(e, CQS_HIP_ERR_INVALID, "splade: null buffer");
    B_TRY(e, hipSetDevice(e->device));
    BSlot* sl; BCtx* c;
    wait_reserved(e, lk);                                     // (another thread's blocking call may hold it: mu is released meanwhile)
    if ((rc = take_slot(e, &sl, &c, true)) != CQS_HIP_OK) return rc;
    uint32_t M = 0;
    rc = splade_forward(e, *c, *sl, tokens, lens, batch, &M);
    if (rc != CQS_HIP_OK) return rc;
    const size_t V = e->cfg.vocab_size;
    if (M == 0) { memset(out_dense, 0, (size_t)batch * V * 4); return CQS_HIP_OK; }   // all empty: ln(1 + 0) = 0 everywhere
    B_TRY(e, hipMemcpyAsync(out_dense, c->dense, (size_t)batch * V * 4, hipMemcpyDeviceToHost, c->stream));
    B_TRY(e, hipStreamSynchronize(c->stream));
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

// The same with the threshold filter of src/splade/mod.rs:1049-1062 on the device, as a TICKET: submit packs the batch
// into a free slot's pinned tables, enqueues H2D + encoder + head + filter + D2H on one of the two execution contexts
// and returns without waiting; collect waits for that ticket and hands out sequence b's entries > threshold as
// (id, weight), ascending id, at out_ids / out_weights [b * cap ..], out_counts[b] = how many passed.  A count above
// `cap` means the row was cut off after its first `cap` entries: the caller re-encodes that sequence through
// cqs_hip_splade_encode (trained models keep 100-300 entries, src/splade/mod.rs:44).  Up to 3 tickets in flight; a
// ticket is released only by collecting it (out_ids = NULL: abandon).
int32_t cqs_hip_splade_submit_sparse(cqs_hip_bert* e, const int32_t* tokens, const uint32_t* lens, uint32_t batch, float threshold,
                                     uint32_t cap, uint64_t* ticket) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_splade_submit_sparse");
    if (!e) return CQS_HIP_ERR_INVALID;
    std::unique_lock<std::mutex> lk(e->mu);
    int32_t rc = check_ready(e, CQS_HIP_BERT_HEAD_MLM);
    if (rc != CQS_HIP_OK) return rc;
    if (!ticket) return bfail(e, CQS_HIP_ERR_INVALID, "splade: null ticket");
    *ticket = 0;
    if (batch == 0 || !lens || cap == 0) return bfail(e, CQS_HIP_ERR_INVALID, "splade: empty batch / null buffer / zero cap");
    B_TRY(e, hipSetDevice(e->device));
    BSlot* sl; BCtx* c;
    const bool reserved = tl_blocking_call;                   // the blocking form (submit + collect) takes the reserved slot
    if (reserved) wait_reserved(e, lk);
    if ((rc = take_slot(e, &sl, &c, reserved)) != CQS_HIP_OK) return rc;
    uint32_t M = 0;
    rc = splade_forward(e, *c, *sl, tokens, lens, batch, &M);
    if (rc != CQS_HIP_OK) return rc;
    const size_t n = (size_t)batch * cap;
    if ((rc = slot_out_reserve(e, *sl, n * 8 + (size_t)batch * 4)) != CQS_HIP_OK) return rc;
    hipStream_t st = c->stream;
    if (M == 0) {
        memset((char*)sl->out + n * 8, 0, (size_t)batch * 4);                      // activations all 0: nothing passes `>`
    } else {
        if ((size_t)batch > c->sp_rows || (size_t)cap > c->sp_cap) {
            B_TRY(e, hipStreamSynchronize(st));
            (void)hipFree(c->sp_ids); (void)hipFree(c->sp_w); (void)hipFree(c->sp_cnt);
            c->sp_ids = c->sp_cnt = nullptr; c->sp_w = nullptr;
            const size_t rows = std::max<size_t>(batch, c->sp_rows), cp = std::max<size_t>(cap, c->sp_cap);
            c->sp_rows = c->sp_cap = 0;
            B_TRY(e, hipMalloc((void**)&c->sp_ids, rows * cp * 4));
            B_TRY(e, hipMalloc((void**)&c->sp_w, rows * cp * 4));
            B_TRY(e, hipMalloc((void**)&c->sp_cnt, rows * 4));
            c->sp_rows = rows; c->sp_cap = cp;
        }
        B_TRY(e, cqs::launch_splade_sparsify(c->dense, batch, e->cfg.vocab_size, threshold, cap, c->sp_ids, c->sp_w, c->sp_cnt, st));
        B_TRY(e, hipMemcpyAsync(sl->out, c->sp_ids, n * 4, hipMemcpyDeviceToHost, st));
        B_TRY(e, hipMemcpyAsync((char*)sl->out + n * 4, c->sp_w, n * 4, hipMemcpyDeviceToHost, st));
        B_TRY(e, hipMemcpyAsync((char*)sl->out + n * 8, c->sp_cnt, (size_t)batch * 4, hipMemcpyDeviceToHost, st));
    }
    B_TRY(e, hipEventRecord(sl->done, st));
    sl->kind = 1; sl->cap = cap;
    sl->ticket = e->next_ticket++;
    *ticket = sl->ticket;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

int32_t cqs_hip_splade_collect_sparse(cqs_hip_bert* e, uint64_t ticket, uint32_t* out_ids, float* out_weights, uint32_t* out_counts) CQS_ABI_TRY {
    if (!e) return CQS_HIP_ERR_INVALID;
    BSlot* sl = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        sl = find_ticket(e, ticket, 1);
        if (!sl) return bfail(e, CQS_HIP_ERR_INVALID, "splade collect: unknown ticket");
        if (sl->collecting) return bfail(e, CQS_HIP_ERR_INVALID, "splade collect: this ticket is already being collected");
        sl->collecting = true;
    }
    struct Release {                                         // whatever path leaves: the slot is free again and waiters hear of it
        cqs_hip_bert* e; BSlot* sl;
        ~Release() { sl->collecting = false; sl->ticket = 0; e->slot_cv.notify_all(); }
    };
    (void)hipSetDevice(e->device);
    const hipError_t he = hipEventSynchronize(sl->done);     // outside the lock: other threads may submit meanwhile
    std::lock_guard<std::mutex> lk(e->mu);
    Release release{e, sl};
    if (he != hipSuccess) return bfail(e, CQS_HIP_ERR_DEVICE, "splade collect: device failure", he);
    if (out_ids) {                                           // NULL: abandon the ticket
        if (!out_weights || !out_counts) return bfail(e, CQS_HIP_ERR_INVALID, "splade collect: null buffer");
        const size_t n = (size_t)sl->B * sl->cap;
        memcpy(out_ids, sl->out, n * 4);
        memcpy(out_weights, (char*)sl->out + n * 4, n * 4);
        memcpy(out_counts, (char*)sl->out + n * 8, (size_t)sl->B * 4);
    }
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

// blocking form: submit + collect
int32_t cqs_hip_splade_encode_sparse(cqs_hip_bert* e, const int32_t* tokens, const uint32_t* lens, uint32_t batch,
                                     float threshold, uint32_t cap, uint32_t* out_ids, float* out_weights, uint32_t* out_counts) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_splade_encode_sparse");
    if (!e) return CQS_HIP_ERR_INVALID;
    if (batch == 0) {
        std::lock_guard<std::mutex> lk(e->mu);
        return check_ready(e, CQS_HIP_BERT_HEAD_MLM);
    }
    if (!out_ids || !out_weights || !out_counts) { std::lock_guard<std::mutex> lk(e->mu); return bfail(e, CQS_HIP_ERR_INVALID, "splade: null buffer / zero cap"); }
    uint64_t t = 0;
    BlockingScope reserved_slot;
    const int32_t rc = cqs_hip_splade_submit_sparse(e, tokens, lens, batch, threshold, cap, &t);
    if (rc != CQS_HIP_OK) return rc;
    return cqs_hip_splade_collect_sparse(e, t, out_ids, out_weights, out_counts);
} CQS_ABI_CATCH(e)

// `compute_scores_opt` below the tokenizer (src/reranker.rs:343-533): (query, passage) pairs already encoded as ids +
// token type ids; out_logits [batch, num_labels] f32 (the caller applies sigmoid to column 0).
int32_t cqs_hip_rerank_logits(cqs_hip_bert* e, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens,
                              uint32_t batch, float* out_logits) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_rerank_logits");
    if (!e) return CQS_HIP_ERR_INVALID;
    std::unique_lock<std::mutex> lk(e->mu);
    int32_t rc = check_ready(e, CQS_HIP_BERT_HEAD_CLASSIFIER);
    if (rc != CQS_HIP_OK) return rc;
    if (batch == 0) return CQS_HIP_OK;
    if (!lens || !out_logits || !tokens) return bfail(e, CQS_HIP_ERR_INVALID, "rerank: null buffer");
    for (uint32_t b = 0; b < batch; ++b)
        if (lens[b] == 0) return bfail(e, CQS_HIP_ERR_INVALID, "rerank: empty sequence (no [CLS] row to pool)");
    B_TRY(e, hipSetDevice(e->device));
    BSlot* sl; BCtx* c;
    wait_reserved(e, lk);                                     // (another thread's blocking call may hold it: mu is released meanwhile)
    if ((rc = take_slot(e, &sl, &c, true)) != CQS_HIP_OK) return rc;
    uint32_t M = 0;
    rc = run_encoder(e, *c, *sl, tokens, type_ids, lens, batch, &M);
    if (rc != CQS_HIP_OK) return rc;
    const cqs_hip_bert_config& cf = e->cfg;
    hipStream_t st = c->stream;
    const uint32_t H = cf.hidden;
    // BertPooler on every sequence's first token (row seq_start[b] of the packed hidden states: the skinny GEMM reads its
    // rows through that table), dense + tanh, then the classifier (labels padded to 16).
    const int32_t* d_start = c->d_meta + (size_t)3 * M;
    B_TRY(e, cqs::launch_gemm_rows(c->x, H, e->wp, e->bp, 1, c->att, batch, H, H, H, cqs::GEMM_OUT_BF16, st, d_start));
    B_TRY(e, cqs::launch_gemm_rows(c->att, H, e->wc, e->bc, 0, c->cls, batch, 16, H, 16, cqs::GEMM_OUT_F32, st));
    std::vector<float> tmp((size_t)batch * 16);
    B_TRY(e, hipMemcpyAsync(tmp.data(), c->cls, tmp.size() * 4, hipMemcpyDeviceToHost, st));
    B_TRY(e, hipStreamSynchronize(st));
    for (uint32_t b = 0; b < batch; ++b)
        for (uint32_t j = 0; j < cf.num_labels; ++j) out_logits[(size_t)b * cf.num_labels + j] = tmp[(size_t)b * 16 + j];
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

// The BERT-family EMBEDDER presets (e5-base, v9-200k, bge-large, bge-large-ft: src/embedder/models.rs:346-405) below the
// tokenizer: `session.run` -> last_hidden_state -> `mean_pool` / `cls_pool` (src/embedder/pooling.rs:87-128), all on the
// device.  out [batch, hidden] f32, NOT normalised (the caller's `normalize_l2`, core.rs:1196-1203).  Ticket form (what
// the index pipeline's embed stage uses, src/cli/pipeline/embedding.rs:226-421) + the blocking `session.run` form.
int32_t cqs_hip_bert_embed_submit(cqs_hip_bert* e, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens, uint32_t batch,
                                  uint32_t pooling, uint64_t* ticket) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_bert_embed_submit");
    if (!e) return CQS_HIP_ERR_INVALID;
    std::unique_lock<std::mutex> lk(e->mu);
    int32_t rc = check_ready(e, CQS_HIP_BERT_HEAD_NONE);
    if (rc != CQS_HIP_OK) return rc;
    if (!ticket) return bfail(e, CQS_HIP_ERR_INVALID, "bert_embed: null ticket");
    *ticket = 0;
    if (batch == 0 || !lens || pooling > 1u) return bfail(e, CQS_HIP_ERR_INVALID, "bert_embed: empty batch / null buffer / unknown pooling");
    {
        uint64_t tot = 0;
        for (uint32_t b = 0; b < batch; ++b) tot += lens[b];
        if (tot && !tokens) return bfail(e, CQS_HIP_ERR_INVALID, "bert_embed: null tokens");
    }
    B_TRY(e, hipSetDevice(e->device));
    BSlot* sl; BCtx* c;
    const bool reserved = tl_blocking_call;                   // the blocking form (submit + collect) takes the reserved slot
    if (reserved) wait_reserved(e, lk);
    if ((rc = take_slot(e, &sl, &c, reserved)) != CQS_HIP_OK) return rc;
    uint32_t M = 0;
    rc = run_encoder(e, *c, *sl, tokens, type_ids, lens, batch, &M);
    if (rc != CQS_HIP_OK) return rc;
    const uint32_t H = e->cfg.hidden;
    if ((rc = slot_out_reserve(e, *sl, (size_t)batch * H * 4)) != CQS_HIP_OK) return rc;
    hipStream_t st = c->stream;
    if (M == 0) {
        memset(sl->out, 0, (size_t)batch * H * 4);                                   // every sequence empty: zero vectors
    } else {
        const int32_t *d_start = c->d_meta + (size_t)3 * M, *d_len = d_start + batch;
        B_TRY(e, cqs::launch_bert_pool(c->x, d_start, d_len, c->dense, batch, H, (int)pooling, st));
        B_TRY(e, hipMemcpyAsync(sl->out, c->dense, (size_t)batch * H * 4, hipMemcpyDeviceToHost, st));
    }
    B_TRY(e, hipEventRecord(sl->done, st));
    sl->kind = 2;
    sl->ticket = e->next_ticket++;
    *ticket = sl->ticket;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

int32_t cqs_hip_bert_embed_collect(cqs_hip_bert* e, uint64_t ticket, float* out) CQS_ABI_TRY {
    if (!e) return CQS_HIP_ERR_INVALID;
    BSlot* sl = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        sl = find_ticket(e, ticket, 2);
        if (!sl) return bfail(e, CQS_HIP_ERR_INVALID, "bert_embed collect: unknown ticket");
        if (sl->collecting) return bfail(e, CQS_HIP_ERR_INVALID, "bert_embed collect: this ticket is already being collected");
        sl->collecting = true;
    }
    struct Release {
        cqs_hip_bert* e; BSlot* sl;
        ~Release() { sl->collecting = false; sl->ticket = 0; e->slot_cv.notify_all(); }
    };
    (void)hipSetDevice(e->device);
    const hipError_t he = hipEventSynchronize(sl->done);
    std::lock_guard<std::mutex> lk(e->mu);
    Release release{e, sl};
    if (he != hipSuccess) return bfail(e, CQS_HIP_ERR_DEVICE, "bert_embed collect: device failure", he);
    if (out) memcpy(out, sl->out, (size_t)sl->B * e->cfg.hidden * 4);               // NULL: abandon the ticket
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

int32_t cqs_hip_bert_embed(cqs_hip_bert* e, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens, uint32_t batch,
                           uint32_t pooling, float* out) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_bert_embed");
    if (!e) return CQS_HIP_ERR_INVALID;
    if (batch == 0) {
        std::lock_guard<std::mutex> lk(e->mu);
        return check_ready(e, CQS_HIP_BERT_HEAD_NONE);
    }
    if (!out) { std::lock_guard<std::mutex> lk(e->mu); return bfail(e, CQS_HIP_ERR_INVALID, "bert_embed: null buffer / unknown pooling"); }
    uint64_t t = 0;
    BlockingScope reserved_slot;
    const int32_t rc = cqs_hip_bert_embed_submit(e, tokens, type_ids, lens, batch, pooling, &t);
    if (rc != CQS_HIP_OK) return rc;
    return cqs_hip_bert_embed_collect(e, t, out);
} CQS_ABI_CATCH(e)

// Diagnostic: final hidden states of the packed tokens, f32 [sum(lens), hidden].
int32_t cqs_hip_bert_hidden(cqs_hip_bert* e, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens,
                            uint32_t batch, float* out_hidden) CQS_ABI_TRY {
    if (!e) return CQS_HIP_ERR_INVALID;
    std::unique_lock<std::mutex> lk(e->mu);
    int32_t rc = check_ready(e, 0xFFFFFFFFu);
    if (rc != CQS_HIP_OK) return rc;
    if (batch == 0) return CQS_HIP_OK;
    if (!lens || !out_hidden || !tokens) return bfail(e, CQS_HIP_ERR_INVALID, "bert: null buffer");
    B_TRY(e, hipSetDevice(e->device));
    BSlot* sl; BCtx* c;
    wait_reserved(e, lk);                                     // (another thread's blocking call may hold it: mu is released meanwhile)
    if ((rc = take_slot(e, &sl, &c, true)) != CQS_HIP_OK) return rc;
    uint32_t M = 0;
    rc = run_encoder(e, *c, *sl, tokens, type_ids, lens, batch, &M);
    if (rc != CQS_HIP_OK || M == 0) return rc;
    std::vector<uint16_t> tmp((size_t)M * e->cfg.hidden);
    B_TRY(e, hipMemcpyAsync(tmp.data(), c->x, tmp.size() * 2, hipMemcpyDeviceToHost, c->stream));
    B_TRY(e, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < tmp.size(); ++i) {
        const uint32_t u = (uint32_t)tmp[i] << 16;
        memcpy(&out_hidden[i], &u, 4);
    }
    return CQS_HIP_OK;
} CQS_ABI_CATCH(e)

// `create_session` for these two models (src/embedder/provider.rs:254-447 via src/splade/mod.rs:433-560 and
// src/reranker.rs:266-330): the local bundle is `{dir}/onnx/model.onnx` (+ external data) or `{dir}/model.onnx`
// (src/reranker.rs:548-556); a Hugging Face checkpoint directory (`model.safetensors`) is accepted as well.
// ONNX initialisers are found as in the embedding engine (onnx_reader.cpp): named tensors by name, anonymous
// transposed MatMul operands through the consuming node's module path.
int32_t cqs_hip_bert_load_dir(const char* model_dir, const cqs_hip_bert_config* cfg, int32_t device, cqs_hip_bert** out) CQS_ABI_TRY {
    if (!model_dir || !cfg || !out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    cqs_hip_bert* e = nullptr;
    int32_t rc = cqs_hip_bert_create(cfg, device, &e);
    if (rc != CQS_HIP_OK) return rc;
    struct Owner {   // every exit but the last, exceptions out of the readers included, gives the handle back
        cqs_hip_bert* p;
        ~Owner() { if (p) cqs_hip_bert_destroy(p); }
    } owner{e};
    const std::string d(model_dir);
    auto exists = [](const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; };
    // names both readers may hand over -> the engine's; unknown tensors (graph constants, position_ids) are skipped
    auto canon = [&](std::string n) -> std::string {
        if (n.rfind("bert.", 0) == 0) n = n.substr(5);
        if (n.rfind("layer.", 0) == 0) n = "encoder." + n;                 // (onnx_reader strips a leading `encoder.`)
        if (n == "cls.predictions.decoder.bias") n = "cls.predictions.bias";
        if (n == "cls.predictions.decoder.weight") return std::string();    // tied to the word embeddings
        const bool known = n.rfind("embeddings.", 0) == 0 || n.rfind("encoder.layer.", 0) == 0 ||
                           (e->cfg.head == CQS_HIP_BERT_HEAD_MLM && n.rfind("cls.predictions.", 0) == 0) ||
                           (e->cfg.head == CQS_HIP_BERT_HEAD_CLASSIFIER && (n.rfind("pooler.", 0) == 0 || n.rfind("classifier.", 0) == 0));
        return known ? n : std::string();
    };
    std::string err;
    int fed = -1;
    std::string onnx = d + "/onnx/model.onnx";
    if (!exists(onnx)) onnx = d + "/model.onnx";
    if (exists(onnx)) {
        fed = cqs_onnx::load(onnx, e->cfg.hidden, 0,
            [&](const std::string& name, const float* data, uint64_t count, const std::vector<uint64_t>&) -> int {
                const std::string n = canon(name);
                if (n.empty()) return 0;
                return cqs_hip_bert_set_tensor(e, n.c_str(), data, count) == CQS_HIP_OK ? 1 : -1;
            }, err);
    } else if (exists(d + "/model.safetensors")) {
        fed = cqs_st::load(d + "/model.safetensors", [&](const std::string& name, const float* data, uint64_t count) -> int {
            const std::string n = canon(name);
            if (n.empty()) return 0;
            return cqs_hip_bert_set_tensor(e, n.c_str(), data, count) == CQS_HIP_OK ? 1 : -1;
        }, err);
    } else {
        err = "no onnx/model.onnx, model.onnx or model.safetensors under " + d;
    }
    if (fed < 0) {
        fprintf(stderr, "[cqs_hip] bert load_dir failed: %s\n", err.c_str());
        return CQS_HIP_ERR_INVALID;
    }
    rc = cqs_hip_bert_finalize(e);
    if (rc != CQS_HIP_OK) {
        fprintf(stderr, "[cqs_hip] bert load_dir failed: %s\n", e->last_error.c_str());
        return rc;
    }
    owner.p = nullptr;
    *out = e;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

void cqs_hip_bert_destroy(cqs_hip_bert* e) CQS_ABI_TRY {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (cqs_hip_bert::Ctx& c : e->ctx)
        if (c.stream) (void)hipStreamSynchronize(c.stream);
    for (void* p : e->owned) (void)hipFree(p);
    for (cqs_hip_bert::Ctx& c : e->ctx) {
        free_scratch(c);
        if (c.stream) (void)hipStreamDestroy(c.stream);
    }
    for (cqs_hip_bert::Slot& sl : e->slot) {
        if (sl.meta) (void)hipHostFree(sl.meta);
        if (sl.out) (void)hipHostFree(sl.out);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    delete e;
} CQS_ABI_CATCH_VOID

uint32_t cqs_hip_bert_vocab(const cqs_hip_bert* e) CQS_ABI_TRY { return e ? e->cfg.vocab_size : 0; } CQS_ABI_CATCH_VAL(0)
int32_t cqs_hip_bert_poisoned(const cqs_hip_bert* e) CQS_ABI_TRY { return e && e->poisoned.load(std::memory_order_acquire) ? 1 : 0; } CQS_ABI_CATCH_NOHANDLE
size_t cqs_hip_bert_last_error(cqs_hip_bert* e, char* buf, size_t cap) CQS_ABI_TRY {
    if (!e || !buf || cap == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    const size_t m = e->last_error.size() < cap - 1 ? e->last_error.size() : cap - 1;
    memcpy(buf, e->last_error.data(), m);
    buf[m] = 0;
    return m;
} CQS_ABI_CATCH_VAL(0)

}  // extern "C"
