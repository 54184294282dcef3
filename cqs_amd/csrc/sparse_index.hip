// sparse_index.hip — the SPLADE sparse retrieval leg on gfx950: `SpladeIndex` (src/splade/index.rs:177-306) behind the C ABI
// (include/cqs_hip.h, "sparse index" section).
//
// The reference keeps `postings: HashMap<u32, Vec<(usize, f32)>>` (token -> [(chunk_index, weight)], chunk order) and, per
// query, walks the posting list of every query term IN QUERY ORDER adding `query_weight * doc_weight` into a
// `HashMap<usize, f32>` (index.rs:248-258), then pushes every scored chunk through `BoundedScoreHeap` (:265-281).
//
// HBM layout here: ONE array of 8-byte postings {chunk, weight bits}, the lists of all tokens back to back, each list in
// ascending chunk order (= the reference's push order); the token -> (start, length) table stays on the host (a query
// names a few dozen tokens).  "chunk" is the chunk's RANK in ascending id order when the caller gives `id_rank`, so that
// the select's (score desc, position asc) order is BoundedScoreHeap's (score desc, id asc).
//
// One launch scores a query (sparse_accumulate_kernel; its comment has the details): every WAVE owns a contiguous range of
// chunks with their scores in LDS, finds its slice of each query term's list in the list's range directory (built at
// create: where the list crosses every range boundary) and streams the slices term after term, 64 postings per step, adding
// `s = s + qw * dw` with separate f32 multiply and add - every chunk's sum is built in exactly the reference's order, so
// the scores are bit-identical, not "close".  No global atomics, no barriers.  Bound by the touched postings' bytes (8 B
// each) + the score row (4 B per chunk); the exact top-k is the dense index's select_finish_kernel over that score row and
// its 64-chunk maxima, with bins laid over the row's own range of maxima (scan_kernels.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "abi_guard.h"
#include "persist_util.h"
#include "scan_kernels.h"

namespace cqs {

constexpr uint32_t kSparsePad = 1024;          // n_pad granule (a multiple of every wave range)
constexpr uint32_t kUnscored = 0xFFFFFFFFu;    // LDS marker: `scores.entry(chunk)` does not exist yet (a NaN bit pattern: weights that
                                               // carry it are refused at build / search, a sum that lands on it is stored as 0x7FC00000)
constexpr uint32_t kMaxTerms = 1u << 16;        // per query
constexpr uint32_t kSparseMaxBatch = 64;       // queries per cqs_hip_sparse_index_search_batch call

struct SparseTerm {            // one query term, resolved on the host
    unsigned long long start;  // first posting of the token's list
    unsigned long long dir;    // first entry of the list's range directory (kNoDir: none - the wave bisects the list)
    uint32_t len;              // postings in the list
    float w;                   // query weight
};
constexpr unsigned long long kNoDir = ~0ull;
constexpr size_t kQoffBytes = ((kSparseMaxBatch + 1) * 4 + 15) / 16 * 16;   // the offsets' share of the query block (terms follow, 16-byte aligned)

__device__ __forceinline__ uint32_t lower_bound_chunk(const uint2* __restrict__ p, uint32_t a, uint32_t b, uint32_t c) {
    while (a < b) {
        const uint32_t m = (a + b) >> 1;
        if (p[m].x < c) a = m + 1u; else b = m;
    }
    return a;
}

// (Round 4, first two forms of this file: (1) every wave bisecting every term's list and walking the terms one after the
// other - one dependent memory round trip per term and wave, 126 us at 64 terms; (2) a `sparse_slice_kernel` in front that
// walked the touched postings once and wrote where each list crosses the wave ranges into a [range][term] table - 12 us
// for the extra launch, the postings read twice (the table term-major instead: that launch 11.9 -> 10.0 us, the scoring
// launch 11.4 -> 14.2).  Now the crossings are precomputed at build time: the range directory.  Also tried on top of it and
// dropped: fetching the slices of 4 x 64 terms together (200 terms: 35.7 -> 38.2 us; the LDS rounds bound it, not the trips).)

// ---- the scoring launch ------------------------------------------------------------------------------------------------
// A wave owns `rw = 1 << sh` chunks and their scores in LDS.  64 terms at a time: lane t reads its term's slice - two
// adjacent entries of the list's range directory (dir[r] = postings of the list with position < r * rw; built once, at
// create), or two bisections for a list too short to have one - a wave
// prefix sum lays the slices end to end (term order, inside a term posting order = the order the reference adds in), and the
// wave streams that sequence 64 postings per step - every lane finds its term by a search over the 64 running counts in
// LDS, so all lanes carry a posting whatever the slices' lengths, and the loads of one step do not wait for the sums of the
// step before.  Two postings of a step may name the same chunk (different terms, or a token a document lists twice): each
// pending lane posts its lane number into claim[chunk] with an LDS minimum, the lowest lane (= the earlier posting) adds
// and clears the claim, the others go round again.  No barriers, no global atomics.
constexpr int kSpUnroll = 4;
// dynamic LDS per wave: rw scores + rw claims + 64 x (count, address lo, address hi, weight)
__host__ __device__ constexpr uint32_t sparse_wave_lds_words(uint32_t rw) { return 2u * rw + 4u * 64u; }

__global__ __launch_bounds__(256) void sparse_accumulate_kernel(const uint2* __restrict__ post, const SparseTerm* __restrict__ terms_all,
                                                                const uint32_t* __restrict__ q_term_off /*[queries + 1]*/, const uint32_t* __restrict__ dir,
                                                                uint32_t n, uint32_t n_pad, uint32_t sh,
                                                                const uint32_t* __restrict__ keep, const uint32_t* __restrict__ chunk_of_rank,
                                                                float* __restrict__ scores, float* __restrict__ gmax,
                                                                uint64_t* __restrict__ gaux, uint32_t group16) {
    extern __shared__ uint32_t sp_lds[];
    // blockIdx.y = the query of a batch: its terms, its score row, its maxima
    const SparseTerm* const terms = terms_all + q_term_off[blockIdx.y];
    const uint32_t n_terms = q_term_off[blockIdx.y + 1u] - q_term_off[blockIdx.y];
    scores += (size_t)blockIdx.y * n_pad;
    gmax += (size_t)blockIdx.y * (group16 ? n_pad >> 4 : n_pad >> 6);
    gaux += (size_t)blockIdx.y * (group16 ? n_pad >> 4 : n_pad >> 6);
    const uint32_t rw = 1u << sh;
    const int lane = threadIdx.x & 63;
    const uint32_t wid = threadIdx.x >> 6;
    const uint32_t range = blockIdx.x * 4u + wid;
    const uint32_t c0 = range << sh;
    if (c0 >= n_pad) return;                              // (wave-uniform; no barrier anywhere in this kernel)
    uint32_t* const my = sp_lds + wid * sparse_wave_lds_words(rw);
    uint32_t* const claim = my + rw;
    uint32_t* const t_cum = claim + rw;                   // [64] postings of this group's terms before term j (in this range)
    uint32_t* const t_alo = t_cum + 64;                   // [64] address of the slice's first posting, low / high word
    uint32_t* const t_ahi = t_alo + 64;
    uint32_t* const t_w = t_ahi + 64;
    for (uint32_t i = lane; i < rw; i += 64u) { my[i] = kUnscored; claim[i] = 0xFFFFFFFFu; }

    for (uint32_t t0 = 0; t0 < n_terms; t0 += 64u) {
        uint32_t len = 0u;
        unsigned long long addr = 0ull;
        float w = 0.f;
        if (t0 + (uint32_t)lane < n_terms) {
            const SparseTerm tm = terms[t0 + lane];
            uint32_t lo, hi;
            if (tm.dir != kNoDir) {
                lo = dir[tm.dir + range];
                hi = dir[tm.dir + range + 1u];
            } else {
                const uint2* const p = post + tm.start;
                lo = lower_bound_chunk(p, 0u, tm.len, c0);
                uint32_t b = tm.len;
                if (lo + rw < b && p[lo + rw].x >= c0 + rw) b = lo + rw;   // the usual case: no more than one posting per chunk
                hi = lower_bound_chunk(p, lo, b, c0 + rw);
            }
            len = hi - lo;
            addr = tm.start + lo;
            w = tm.w;
        }
        uint32_t cum = len;                               // inclusive wave prefix sum
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(cum, off, 64);
            if (lane >= off) cum += v;
        }
        const uint32_t total = __builtin_amdgcn_readlane(cum, 63);
        if (total == 0u) continue;
        t_cum[lane] = cum - len;
        t_alo[lane] = (uint32_t)addr;
        t_ahi[lane] = (uint32_t)(addr >> 32);
        t_w[lane] = __builtin_bit_cast(uint32_t, w);
        for (uint32_t g0 = 0; g0 < total; g0 += 64u * kSpUnroll) {
            uint2 e[kSpUnroll];
            float prod[kSpUnroll];
            bool pend[kSpUnroll];
#pragma unroll
            for (int u = 0; u < kSpUnroll; ++u) {
                const uint32_t g = g0 + 64u * u + (uint32_t)lane;
                pend[u] = g < total;
                e[u] = make_uint2(c0, 0u);
                prod[u] = 0.f;
                if (pend[u]) {
                    uint32_t a = 0u, b = 64u;             // the last term whose running count <= g
#pragma unroll
                    for (int step = 0; step < 6; ++step) {
                        const uint32_t m = (a + b) >> 1;
                        if (t_cum[m] <= g) a = m; else b = m;
                    }
                    const unsigned long long at = ((unsigned long long)t_ahi[a] << 32 | t_alo[a]) + (g - t_cum[a]);
#if defined(CQS_SP_ABLATE_NOLOAD)      // timing experiment (wrong results): no posting is read
                    e[u] = make_uint2(c0 + ((uint32_t)at & (rw - 1u)), 0x3f800000u);
#else
                    e[u] = post[at];
#endif
                    prod[u] = __fmul_rn(__builtin_bit_cast(float, t_w[a]), __builtin_bit_cast(float, e[u].y));
                }
            }
#pragma unroll
            for (int u = 0; u < kSpUnroll; ++u) {
                if (g0 + 64u * u >= total) break;         // (wave-uniform)
                const uint32_t at = e[u].x - c0;
                bool pending = pend[u];
#if defined(CQS_SP_ABLATE_NORMW)       // timing experiment (wrong results): the loads alone
                if (pending && prod[u] == 12345.678f) my[at] = 1u;
                pending = false;
#endif
                while (__builtin_amdgcn_ballot_w64(pending) != 0ull) {
                    // (compiler barriers: the other lanes' LDS stores are invisible to the single-thread view the optimiser
                    // reasons in - no load of a score or a claim may move across a round)
                    asm volatile("" ::: "memory");
                    if (pending) atomicMin(&claim[at], (uint32_t)lane);
                    asm volatile("" ::: "memory");
                    if (pending && *(volatile uint32_t*)&claim[at] == (uint32_t)lane) {
                        const uint32_t s = *(volatile uint32_t*)&my[at];
                        const uint32_t sum = __builtin_bit_cast(uint32_t, __fadd_rn(s == kUnscored ? 0.f : __builtin_bit_cast(float, s), prod[u]));
                        // a NaN sum that happens to carry the marker's bits (the sign of a NaN result is the hardware's
                        // choice) is stored as the canonical NaN: still NaN for every later add, still dropped at the end
                        *(volatile uint32_t*)&my[at] = sum == kUnscored ? 0x7FC00000u : sum;
                        *(volatile uint32_t*)&claim[at] = 0xFFFFFFFFu;
                        pending = false;
                    }
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    // the score row + the maxima of its 64-chunk groups (what select_finish_kernel reads); a chunk that was never scored,
    // is filtered out, or whose score is not finite (BoundedScoreHeap refuses it, candidate.rs:245-247) = -inf
    for (uint32_t i = lane; i < rw; i += 64u) {
        const uint32_t r = c0 + i;
        const uint32_t s = my[i];
        float v = __builtin_bit_cast(float, s);
        bool ok = r < n && s != kUnscored && __builtin_isfinite(v);
        if (ok && keep) {
            const uint32_t c = chunk_of_rank ? chunk_of_rank[r] : r;
            ok = (keep[c >> 5] >> (c & 31u)) & 1u;
        }
        v = ok ? v : -INFINITY;
        scores[r] = v;
        float m = cqs::wave_max16(v);                     // (DPP: `__shfl_xor` is a ds_bpermute round trip per step on gfx950)
        // beside each maximum: the lane it sits in and the group's runner-up (select_finish_kernel, round 5: a group whose
        // runner-up misses the threshold contributes its maximum without its scores being read back)
        if (group16) {                                    // small indexes: maxima of 16 chunks (fewer than k groups of 64 would
            if ((lane & 15) == 0) gmax[r >> 4] = m;       // make every score a candidate and send the select down its radix path)
            const uint32_t seg = (uint32_t)(__ballot(v == m) >> (lane & 48)) & 0xFFFFu;   // my 16 lanes (no scored chunk: m = -inf, all match)
            const uint32_t arg = (uint32_t)__builtin_ctz(seg);
            const float sec = cqs::wave_max16(((uint32_t)(lane & 15) == arg) ? -INFINITY : v);
            if ((lane & 15) == 0) gaux[r >> 4] = ((uint64_t)arg << 32) | (uint64_t)__builtin_bit_cast(uint32_t, sec);
        } else {
            m = cqs::wave_max64(v);
            if (lane == 0) gmax[r >> 6] = m;
            const uint32_t arg = (uint32_t)__builtin_ctzll(__ballot(v == m));
            const float sec = cqs::wave_max64(((uint32_t)lane == arg) ? -INFINITY : v);
            if (lane == 0) gaux[r >> 6] = ((uint64_t)arg << 32) | (uint64_t)__builtin_bit_cast(uint32_t, sec);
        }
    }
}

}  // namespace cqs

using cqs::kMaxTerms;
using cqs::kSparseMaxBatch;
using cqs::kQoffBytes;
using cqs::kNoDir;
using cqs::kSparsePad;
using cqs::kUnscored;
using cqs::sparse_accumulate_kernel;
using cqs::sparse_wave_lds_words;
using cqs::SparseTerm;

// One single-query search waiting for a shared pair of launches (the combining queue of cqs_hip_sparse_index_search).
// Lives on its caller's stack; the pointers are the caller's own buffers.
struct cqs_sparse_req {
    const uint32_t* q_tokens;
    const float* q_weights;
    uint32_t n_terms, k;
    uint64_t* out_chunks;
    float* out_scores;
    uint32_t* out_count;
    int32_t rc = 0;
    bool done = false;
};

struct cqs_hip_sparse_index {
    std::mutex mu;
    // combining queue (as the dense index's, index.hip): concurrent unfiltered single-query calls share launches
    std::mutex cmu;
    std::condition_variable ccv;
    std::deque<cqs_sparse_req*> pending;
    bool leader = false;
    bool combine = true;                     // CQS_HIP_COMBINE=0 turns it off (read at create)
    uint32_t combine_wait_us = 100;          // CQS_HIP_COMBINE_WAIT_US
    std::chrono::steady_clock::time_point last_pass_end{};   // guarded by cmu
    uint32_t expect = 1;                     // like-parameter callers recent passes saw
    std::atomic<uint64_t> stat_passes{0}, stat_queries{0};
    std::string last_error;
    std::atomic<bool> poisoned{false};
    int device = 0;
    uint64_t n = 0, n_postings = 0;
    uint32_t n_pad = 0, rw = 64, sh = 6, n_cu = 256;
    bool ranked = false;
    bool group16 = false;                    // maxima per 16 chunks instead of 64 (indexes up to 262 144 chunks)
    std::vector<uint32_t> tok;               // sorted distinct token ids
    std::vector<uint64_t> off;               // [tok.size() + 1]
    std::vector<uint32_t> chunk_of_rank;     // host copy (empty: identity)
    hipStream_t stream = nullptr;
    uint2* d_post = nullptr;
    uint32_t* d_chunk_of_rank = nullptr;
    float* d_scores = nullptr;
    float* d_gmax = nullptr;
    uint32_t* d_work = nullptr;
    uint32_t* d_keep = nullptr;
    SparseTerm* d_terms = nullptr;
    uint32_t* d_qoff = nullptr;              // [kSparseMaxBatch + 1] first term of every query of a batch
    uint32_t* h_qoff = nullptr;              // pinned
    uint32_t b_cap = 0;                      // queries the score / maxima / key scratch holds
    uint32_t terms_cap = 0;
    SparseTerm* h_terms = nullptr;           // pinned
    uint32_t* d_dir = nullptr;               // range directories, list after list: n_pad / rw + 1 entries each
    std::vector<uint64_t> dir_off;           // [tok.size()]: a list's first entry in d_dir, kNoDir = none
    uint64_t dir_entries = 0;
    uint32_t* h_keep = nullptr;              // pinned, ceil(n / 32) words
    uint64_t* d_out_keys = nullptr;
    uint32_t* d_out_count = nullptr;
    uint64_t* h_out_keys = nullptr;          // pinned + device-visible, kMaxK + 1 words (the last one: the count): the select writes here
    uint64_t* h_out_keys_dev = nullptr;      // its device address (null: not mappable -> device buffer + two copies)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long* d_dbg = nullptr;     // CQS_HIP_DEBUG_STAMPS=1: select_finish phase stamps of the last search (printed to stderr)
    float last_ms = 0.f;
    std::atomic<bool> want_timing{false};    // set by the first last_search that asks for the time: searches are timed from then on
    uint64_t last_touched = 0;
};

namespace {

int32_t sfail(cqs_hip_sparse_index* s, int32_t code, const std::string& what, hipError_t he = hipSuccess) {
    s->last_error = what;
    if (he != hipSuccess) s->last_error += std::string(": ") + hipGetErrorString(he);
    if (code == CQS_HIP_ERR_DEVICE) s->poisoned = true;
    return code;
}
#define S_TRY(s, expr)                                                                   \
    do {                                                                                 \
        const hipError_t he_ = (expr);                                                   \
        if (he_ != hipSuccess) return sfail((s), he_ == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE, #expr, he_); \
    } while (0)

void release(cqs_hip_sparse_index* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (void* p : {(void*)s->d_post, (void*)s->d_chunk_of_rank, (void*)s->d_scores, (void*)s->d_gmax, (void*)s->d_work,
                    (void*)s->d_keep, (void*)s->d_qoff, (void*)s->d_dir, (void*)s->d_out_keys, (void*)s->d_out_count, (void*)s->d_dbg})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)s->h_keep, (void*)s->h_out_keys, (void*)s->h_qoff})      // (d_terms / h_terms point into the qoff blocks)
        if (p) (void)hipHostFree(p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int32_t ensure_terms(cqs_hip_sparse_index* s, uint32_t t) {
    if (t <= s->terms_cap) return CQS_HIP_OK;
    const uint32_t cap = std::max(256u, t + t / 2u);
    // ONE block per side: [first-term offsets of the batch's queries: kQoffBytes][terms] - a search is one H2D copy
    // (round 4 sent the offsets and the terms in two)
    if (s->d_qoff) { (void)hipFree(s->d_qoff); s->d_qoff = nullptr; s->d_terms = nullptr; }
    if (s->h_qoff) { (void)hipHostFree(s->h_qoff); s->h_qoff = nullptr; s->h_terms = nullptr; }
    s->terms_cap = 0;
    S_TRY(s, hipMalloc((void**)&s->d_qoff, kQoffBytes + (size_t)cap * sizeof(SparseTerm)));
    S_TRY(s, hipHostMalloc((void**)&s->h_qoff, kQoffBytes + (size_t)cap * sizeof(SparseTerm), hipHostMallocDefault));
    s->d_terms = (SparseTerm*)((char*)s->d_qoff + kQoffBytes);
    s->h_terms = (SparseTerm*)((char*)s->h_qoff + kQoffBytes);
    s->terms_cap = cap;
    return CQS_HIP_OK;
}

// score rows, maxima and result keys for `b` queries (grown on demand; one query's worth exists from create on)
int32_t ensure_batch(cqs_hip_sparse_index* s, uint32_t b) {
    if (b <= s->b_cap) return CQS_HIP_OK;
    for (void** p : {(void**)&s->d_scores, (void**)&s->d_gmax, (void**)&s->d_out_keys, (void**)&s->d_out_count})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (s->h_out_keys) { (void)hipHostFree(s->h_out_keys); s->h_out_keys = nullptr; s->h_out_keys_dev = nullptr; }
    s->b_cap = 0;
    const size_t groups = s->n_pad / (s->group16 ? 16u : 64u);
    S_TRY(s, hipMalloc((void**)&s->d_scores, (size_t)b * s->n_pad * 4));
    S_TRY(s, hipMalloc((void**)&s->d_gmax, (size_t)b * groups * 12));   // maxima, then (argmax, runner-up) pairs (groups % 4 == 0: 8-byte aligned)
    S_TRY(s, hipMalloc((void**)&s->d_out_keys, (size_t)b * cqs::kMaxK * 8));
    S_TRY(s, hipMalloc((void**)&s->d_out_count, (size_t)b * 4));
    // keys [b][k] then the b counts, in one pinned + device-visible block: the select writes there
    S_TRY(s, hipHostMalloc((void**)&s->h_out_keys, (size_t)b * (cqs::kMaxK + 1) * 8, hipHostMallocMapped));
    if (hipHostGetDevicePointer((void**)&s->h_out_keys_dev, s->h_out_keys, 0) != hipSuccess) s->h_out_keys_dev = nullptr;
    s->b_cap = b;
    return CQS_HIP_OK;
}

// Second half of every constructor: `s->tok` / `s->off` and the host posting array are final; device, wave ranges, range
// directories, scratch.  On failure the caller's guard releases the handle.
int32_t finish_create(cqs_hip_sparse_index* s, const std::vector<uint2>& post, int32_t device) {
    const uint64_t n = s->n, P = s->n_postings;
    auto dfail = [&](hipError_t he) -> int32_t {           // (the caller's guard releases)
        return he == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE;
    };
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return dfail(he);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cu = (uint32_t)prop.multiProcessorCount;
    // chunks per wave: enough waves to keep ~16 per CU in flight, 64 ... 1024 chunks each
    s->rw = 1024;
    while (s->rw > 64u && s->n_pad / s->rw < s->n_cu * 16u) s->rw >>= 1;
    s->sh = 0;
    while ((1u << s->sh) < s->rw) ++s->sh;
    // Range directories: dir[r] = how many postings of the list sit below position r * rw, r = 0 .. n_pad / rw - what a wave
    // needs to find its slice of the list with two adjacent loads.  Longest lists first, within a budget of entries equal
    // to the postings' own bytes (2 P entries x 4 B = P x 8 B; never less than 64 MB.  Round 4's floor was 256 MB: a 100k-chunk
    // index with a 30k-token vocabulary paid 192 MB of directories - and the same again as a host staging vector - for
    // 80 MB of postings): at 1M chunks rw = 128, so a list's directory is 7 813 + 1 entries and the ~25 k longest
    // lists of a 96 M-posting index get one; lists left without (shorter than kDirMinLen, or past the budget) are bisected.
    {
        constexpr uint64_t kDirMinLen = 32;
        const uint64_t R1 = (uint64_t)(s->n_pad / s->rw) + 1;
        const uint64_t budget = std::max<uint64_t>(16ull << 20, 2ull * P);
        s->dir_off.assign(s->tok.size(), kNoDir);
        std::vector<uint32_t> order;
        for (size_t t = 0; t < s->tok.size(); ++t)
            if (s->off[t + 1] - s->off[t] >= kDirMinLen) order.push_back((uint32_t)t);
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const uint64_t lx = s->off[x + 1] - s->off[x], ly = s->off[y + 1] - s->off[y];
            return lx != ly ? lx > ly : x < y;
        });
        uint64_t used = 0;
        for (uint32_t t : order) {
            if (used + R1 > budget || s->off[t + 1] - s->off[t] > 0xFFFFFFFFull) break;
            s->dir_off[t] = used;
            used += R1;
        }
        s->dir_entries = used;
        std::vector<uint32_t> dirs((size_t)used);
        for (size_t t = 0; t < s->tok.size(); ++t) {
            if (s->dir_off[t] == kNoDir) continue;
            const uint2* const pl = post.data() + s->off[t];
            const uint32_t len = (uint32_t)(s->off[t + 1] - s->off[t]);
            uint32_t* const d = dirs.data() + s->dir_off[t];
            uint32_t pos = 0;
            for (uint64_t r = 0; r < R1; ++r) {
                const uint64_t edge = r * s->rw;
                while (pos < len && pl[pos].x < edge) ++pos;
                d[r] = pos;
            }
        }
        if ((he = hipMalloc((void**)&s->d_dir, std::max<size_t>((size_t)used, 1) * 4)) != hipSuccess) return dfail(he);
        if (used && (he = hipMemcpy(s->d_dir, dirs.data(), (size_t)used * 4, hipMemcpyHostToDevice)) != hipSuccess) return dfail(he);
    }
    if ((he = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)) != hipSuccess) return dfail(he);
    if ((he = hipEventCreate(&s->ev0)) != hipSuccess || (he = hipEventCreate(&s->ev1)) != hipSuccess) return dfail(he);
    if ((he = hipMalloc((void**)&s->d_post, std::max<size_t>((size_t)P, 1) * sizeof(uint2))) != hipSuccess) return dfail(he);
    if (P && (he = hipMemcpy(s->d_post, post.data(), (size_t)P * sizeof(uint2), hipMemcpyHostToDevice)) != hipSuccess) return dfail(he);
    if (s->ranked) {
        if ((he = hipMalloc((void**)&s->d_chunk_of_rank, std::max<size_t>((size_t)n, 1) * 4)) != hipSuccess) return dfail(he);
        if (n && (he = hipMemcpy(s->d_chunk_of_rank, s->chunk_of_rank.data(), (size_t)n * 4, hipMemcpyHostToDevice)) != hipSuccess) return dfail(he);
    }
    // measured (select at k = 500): 20k chunks 40 -> 24 us, 100k 29 -> 23, 1M 29 -> 45: 16-chunk groups while their maxima fit
    // one pass of the select's workgroup (1024 threads x 16 registers)
    s->group16 = s->n_pad / 16u <= 16384u;
    if ((he = hipMalloc((void**)&s->d_work, cqs::kWorkWords * 4)) != hipSuccess) return dfail(he);
    if ((he = hipMemset(s->d_work, 0, cqs::kWorkWords * 4)) != hipSuccess) return dfail(he);
    if ((he = hipMalloc((void**)&s->d_keep, (size_t)(s->n_pad / 32u) * 4)) != hipSuccess) return dfail(he);
    if ((he = hipHostMalloc((void**)&s->h_keep, (size_t)(s->n_pad / 32u) * 4, hipHostMallocDefault)) != hipSuccess) return dfail(he);
    {
        const int32_t rc = ensure_batch(s, 1u);
        if (rc != CQS_HIP_OK) return rc;
    }
    if (const char* e = getenv("CQS_HIP_DEBUG_STAMPS"); e && *e == '1')
        if (hipMalloc((void**)&s->d_dbg, 16 * 8) == hipSuccess) (void)hipMemset(s->d_dbg, 0, 16 * 8);
    if (const char* e = getenv("CQS_HIP_COMBINE")) s->combine = !(e[0] == '0');
    if (const char* e = getenv("CQS_HIP_COMBINE_WAIT_US")) s->combine_wait_us = (uint32_t)strtoul(e, nullptr, 10);
    return CQS_HIP_OK;
}

struct CreateGuard {                                       // an exception or an early return frees what exists so far
    cqs_hip_sparse_index* p;
    ~CreateGuard() { if (p) release(p); }
};

// First half: the handle with its chunk count and id order.  nullptr = id_rank is not a permutation.
cqs_hip_sparse_index* begin_create(uint64_t n, const uint32_t* id_rank, int32_t device) {
    cqs_hip_sparse_index* s = new cqs_hip_sparse_index();
    CreateGuard guard{s};
    s->device = device;
    s->n = n;
    s->n_pad = (uint32_t)((n + kSparsePad - 1) / kSparsePad * kSparsePad);
    if (s->n_pad == 0) s->n_pad = kSparsePad;
    if (id_rank) {                                         // chunk_of_rank[r] = the chunk whose id is the r-th smallest
        s->chunk_of_rank.assign((size_t)n, 0xFFFFFFFFu);
        for (uint64_t i = 0; i < n; ++i) {
            if (id_rank[i] >= n || s->chunk_of_rank[id_rank[i]] != 0xFFFFFFFFu) return nullptr;
            s->chunk_of_rank[id_rank[i]] = (uint32_t)i;
        }
        s->ranked = true;
    }
    guard.p = nullptr;
    return s;
}

}  // namespace

extern "C" {

int32_t cqs_hip_sparse_index_create(const uint64_t* doc_off, const uint32_t* tokens, const float* weights, uint64_t n,
                                    const uint32_t* id_rank, int32_t device, cqs_hip_sparse_index** out) CQS_ABI_TRY {
    if (!out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    if (n && !doc_off) return CQS_HIP_ERR_INVALID;
    if (n >= 0xFFFFFFFFull - kSparsePad) return CQS_HIP_ERR_INVALID;
    const uint64_t P = n ? doc_off[n] : 0;
    if (n && doc_off[0] != 0) return CQS_HIP_ERR_INVALID;
    for (uint64_t i = 0; i < n; ++i)
        if (doc_off[i + 1] < doc_off[i]) return CQS_HIP_ERR_INVALID;
    if (P && (!tokens || !weights)) return CQS_HIP_ERR_INVALID;
    for (uint64_t e = 0; e < P; ++e) {
        uint32_t bits;
        memcpy(&bits, &weights[e], 4);
        if (bits == kUnscored) return CQS_HIP_ERR_INVALID;      // the one NaN payload the kernel reserves
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CQS_HIP_ERR_NO_DEVICE;
    cqs_hip_sparse_index* s = begin_create(n, id_rank, device);
    if (!s) return CQS_HIP_ERR_INVALID;                     // id_rank is not a permutation of 0 .. n - 1
    CreateGuard guard{s};
    s->n_postings = P;
    // the token table: sorted distinct ids; a dense counting pass when the ids are small (every real vocabulary), a sort otherwise
    uint32_t max_tok = 0;
    for (uint64_t e = 0; e < P; ++e) max_tok = std::max(max_tok, tokens[e]);
    std::vector<uint64_t> dense;                          // token -> slot + 1 (dense path)
    std::unordered_map<uint32_t, uint32_t> sparse_slot;   // (sort path)
    const bool use_dense = P && max_tok < (1u << 24);
    if (use_dense) {
        dense.assign((size_t)max_tok + 1, 0);
        for (uint64_t e = 0; e < P; ++e) dense[tokens[e]]++;
        for (uint32_t t = 0; t <= max_tok; ++t)
            if (dense[t]) { s->tok.push_back(t); s->off.push_back(dense[t]); }
    } else if (P) {
        std::vector<uint32_t> sorted(tokens, tokens + P);
        std::sort(sorted.begin(), sorted.end());
        for (uint64_t e = 0; e < P;) {
            uint64_t f = e;
            while (f < P && sorted[f] == sorted[e]) ++f;
            s->tok.push_back(sorted[e]);
            s->off.push_back(f - e);
            e = f;
        }
    }
    {   // counts -> offsets
        uint64_t run = 0;
        for (size_t t = 0; t < s->off.size(); ++t) { const uint64_t c = s->off[t]; s->off[t] = run; run += c; }
        s->off.push_back(run);
    }
    if (use_dense) {
        for (size_t t = 0; t < s->tok.size(); ++t) dense[s->tok[t]] = t + 1;
    } else {
        sparse_slot.reserve(s->tok.size());
        for (size_t t = 0; t < s->tok.size(); ++t) sparse_slot[s->tok[t]] = (uint32_t)t;
    }
    // the postings, list by list; inside a list ascending position (rank order = the order the documents are walked in)
    std::vector<uint2> post((size_t)P);
    {
        std::vector<uint64_t> cur(s->off.begin(), s->off.end() - 1);
        for (uint64_t r = 0; r < n; ++r) {
            const uint64_t d = s->ranked ? s->chunk_of_rank[r] : r;
            for (uint64_t e = doc_off[d]; e < doc_off[d + 1]; ++e) {
                const size_t slot = use_dense ? (size_t)(dense[tokens[e]] - 1) : (size_t)sparse_slot[tokens[e]];
                uint32_t bits;
                memcpy(&bits, &weights[e], 4);
                post[cur[slot]++] = make_uint2((uint32_t)r, bits);
            }
        }
    }
    const int32_t rc = finish_create(s, post, device);
    if (rc != CQS_HIP_OK) return rc;
    guard.p = nullptr;
    *out = s;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

int32_t cqs_hip_sparse_index_create_inverted(const uint32_t* token_ids, const uint64_t* list_off, const uint32_t* post_chunks,
                                             const float* post_weights, uint64_t n_tokens, uint64_t n, const uint32_t* id_rank,
                                             int32_t device, cqs_hip_sparse_index** out) CQS_ABI_TRY {
    if (!out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    if (n >= 0xFFFFFFFFull - kSparsePad || n_tokens > 0xFFFFFFFFull) return CQS_HIP_ERR_INVALID;
    if (n_tokens && (!token_ids || !list_off)) return CQS_HIP_ERR_INVALID;
    const uint64_t P_in = n_tokens ? list_off[n_tokens] : 0;
    if (n_tokens && list_off[0] != 0) return CQS_HIP_ERR_INVALID;
    for (uint64_t t = 0; t < n_tokens; ++t)
        if (list_off[t + 1] < list_off[t]) return CQS_HIP_ERR_INVALID;
    if (P_in && (!post_chunks || !post_weights)) return CQS_HIP_ERR_INVALID;
    for (uint64_t e = 0; e < P_in; ++e) {
        uint32_t bits;
        memcpy(&bits, &post_weights[e], 4);
        if (bits == kUnscored) return CQS_HIP_ERR_INVALID;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CQS_HIP_ERR_NO_DEVICE;
    cqs_hip_sparse_index* s = begin_create(n, id_rank, device);
    if (!s) return CQS_HIP_ERR_INVALID;
    CreateGuard guard{s};
    // the HashMap's keys in ascending order (a key given twice is refused: a map has each key once)
    std::vector<uint32_t> order((size_t)n_tokens);
    for (uint64_t t = 0; t < n_tokens; ++t) order[t] = (uint32_t)t;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return token_ids[x] < token_ids[y]; });
    for (uint64_t t = 1; t < n_tokens; ++t)
        if (token_ids[order[t]] == token_ids[order[t - 1]]) return CQS_HIP_ERR_INVALID;
    std::vector<uint32_t> rank_of;                          // chunk -> position
    if (s->ranked) {
        rank_of.resize((size_t)n);
        for (uint64_t r = 0; r < n; ++r) rank_of[s->chunk_of_rank[r]] = (uint32_t)r;
    }
    std::vector<uint2> post;
    post.reserve((size_t)P_in);
    s->off.push_back(0);
    for (uint64_t t = 0; t < n_tokens; ++t) {
        const uint32_t src = order[t];
        const size_t first = post.size();
        bool ascending = true;
        for (uint64_t e = list_off[src]; e < list_off[src + 1]; ++e) {
            const uint32_t c = post_chunks[e];
            if (c >= n) continue;                           // the search skips such a posting (index.rs:252): it never scores
            uint32_t bits;
            memcpy(&bits, &post_weights[e], 4);
            const uint32_t pos = s->ranked ? rank_of[c] : c;
            if (post.size() > first && pos < post.back().x) ascending = false;
            post.push_back(make_uint2(pos, bits));
        }
        if (post.size() == first) continue;                 // nothing usable under this token: no list
        // ascending positions; postings of ONE chunk keep their order (the only order the sums depend on)
        if (!ascending) std::stable_sort(post.begin() + (ptrdiff_t)first, post.end(), [](const uint2& x, const uint2& y) { return x.x < y.x; });
        s->tok.push_back(token_ids[src]);
        s->off.push_back(post.size());
    }
    s->n_postings = post.size();
    const int32_t rc = finish_create(s, post, device);
    if (rc != CQS_HIP_OK) return rc;
    guard.p = nullptr;
    *out = s;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

// ---- persistence (the role of `SpladeIndex::save` / `load`, src/splade/index.rs:346-1072: skip the rebuild after a restart;
// invalidated by the store's `splade_generation` counter).  Own format - the reference's file is not read or written:
// 64-byte header {magic "CQSHIPS1", version, ranked, chunks, tokens, postings, generation, checksum} + four sections,
// each zero-padded to 8 bytes: token ids (u32, ascending), list offsets (u64, tokens + 1), postings ({position, weight
// bits}, 8 B each), and - ranked indexes only - chunk_of_rank (u32 per chunk).  The checksum covers the sections.
namespace {
struct SparseFileHeader {
    char magic[8];
    uint32_t version, ranked;
    uint64_t chunks, tokens, postings, generation, checksum;
    uint8_t reserved[8];
};
static_assert(sizeof(SparseFileHeader) == 64, "header is 64 bytes");
const char kSparseMagic[8] = {'C', 'Q', 'S', 'H', 'I', 'P', 'S', '1'};
inline size_t pad8(size_t b) { return (b + 7) & ~(size_t)7; }
}  // namespace

int32_t cqs_hip_sparse_index_save(cqs_hip_sparse_index* s, const char* path, uint64_t generation, uint64_t* out_checksum) CQS_ABI_TRY {
    if (!s || !path) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->mu);
    if (s->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    S_TRY(s, hipSetDevice(s->device));
    S_TRY(s, hipStreamSynchronize(s->stream));
    std::vector<uint2> post((size_t)s->n_postings);
    if (s->n_postings) S_TRY(s, hipMemcpy(post.data(), s->d_post, (size_t)s->n_postings * sizeof(uint2), hipMemcpyDeviceToHost));
    const size_t U = s->tok.size();
    struct Sec { const void* p; size_t bytes; };
    const Sec secs[4] = {{s->tok.data(), U * 4}, {s->off.data(), (U + 1) * 8}, {post.data(), post.size() * 8},
                         {s->chunk_of_rank.data(), s->ranked ? (size_t)s->n * 4 : 0}};
    size_t total = 0;
    for (const Sec& c : secs) total += pad8(c.bytes);
    SparseFileHeader h{};
    memcpy(h.magic, kSparseMagic, 8);
    h.version = 1;
    h.ranked = s->ranked ? 1u : 0u;
    h.chunks = s->n; h.tokens = U; h.postings = s->n_postings; h.generation = generation;
    cqs_persist::Checksum ck(total);
    const std::string live(path), tmp = live + ".tmp";
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return sfail(s, CQS_HIP_ERR_INVALID, "sparse save: cannot create " + tmp);
    bool ok = cqs_persist::write_all(fd, &h, sizeof h);           // checksum patched in below
    const uint64_t zero = 0;
    for (int i = 0; i < 4 && ok; ++i) {
        const size_t padded = pad8(secs[i].bytes);
        if (padded == secs[i].bytes) {
            ck.update(secs[i].p, secs[i].bytes, false);
            ok = cqs_persist::write_all(fd, secs[i].p, secs[i].bytes);
        } else {                                                  // (only a u32 section of odd length: copy its last word)
            const size_t whole = secs[i].bytes & ~(size_t)7;
            ck.update(secs[i].p, whole, false);
            uint64_t tail = 0;
            memcpy(&tail, (const uint8_t*)secs[i].p + whole, secs[i].bytes - whole);
            ck.update(&tail, 8, false);
            ok = cqs_persist::write_all(fd, secs[i].p, secs[i].bytes) && cqs_persist::write_all(fd, &zero, padded - secs[i].bytes);
        }
    }
    ck.update(nullptr, 0, true);
    h.checksum = ck.finish();
    ok = ok && lseek(fd, 0, SEEK_SET) == 0 && cqs_persist::write_all(fd, &h, sizeof h) && fsync(fd) == 0;
    close(fd);
    if (!ok) { unlink(tmp.c_str()); return sfail(s, CQS_HIP_ERR_INVALID, "sparse save: write failed"); }
    if (rename(tmp.c_str(), live.c_str()) != 0) { unlink(tmp.c_str()); return sfail(s, CQS_HIP_ERR_INVALID, "sparse save: rename failed"); }
    cqs_persist::fsync_parent(live);
    if (out_checksum) *out_checksum = h.checksum;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(s)

int32_t cqs_hip_sparse_index_load(const char* path, uint64_t expected_chunks, uint64_t generation, int32_t device,
                                  cqs_hip_sparse_index** out) CQS_ABI_TRY {
    if (!out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    if (!path) return CQS_HIP_ERR_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CQS_HIP_ERR_NO_DEVICE;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return CQS_HIP_ERR_INVALID;                        // missing file: the caller builds (index.rs:1073-1107)
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    SparseFileHeader h;
    struct stat st;
    if (fstat(fd, &st) != 0 || !cqs_persist::read_all(fd, &h, sizeof h)) return CQS_HIP_ERR_INVALID;
    if (memcmp(h.magic, kSparseMagic, 8) != 0 || h.version != 1 || h.ranked > 1) return CQS_HIP_ERR_INVALID;
    if (h.generation != generation) return CQS_HIP_ERR_INVALID;    // the store changed since the file was written (index.rs:1011-1019)
    if (expected_chunks && h.chunks != expected_chunks) return CQS_HIP_ERR_INVALID;
    if (h.chunks >= 0xFFFFFFFFull - kSparsePad || h.tokens > 0xFFFFFFFFull || h.postings > (1ull << 40)) return CQS_HIP_ERR_INVALID;
    const size_t b_tok = (size_t)h.tokens * 4, b_off = ((size_t)h.tokens + 1) * 8, b_post = (size_t)h.postings * 8,
                 b_rank = h.ranked ? (size_t)h.chunks * 4 : 0;
    const size_t total = pad8(b_tok) + pad8(b_off) + pad8(b_post) + pad8(b_rank);
    if ((uint64_t)st.st_size != sizeof h + total) return CQS_HIP_ERR_INVALID;     // truncated or grown
    cqs_hip_sparse_index* s = new cqs_hip_sparse_index();
    CreateGuard guard{s};
    s->device = device;
    s->n = h.chunks;
    s->n_postings = h.postings;
    s->n_pad = (uint32_t)((h.chunks + kSparsePad - 1) / kSparsePad * kSparsePad);
    if (s->n_pad == 0) s->n_pad = kSparsePad;
    s->ranked = h.ranked != 0;
    std::vector<uint64_t> tokbuf(pad8(b_tok) / 8), rankbuf(pad8(b_rank) / 8);
    s->off.resize((size_t)h.tokens + 1);
    std::vector<uint2> post((size_t)h.postings);
    cqs_persist::Checksum ck(total);
    if (!cqs_persist::read_all(fd, tokbuf.data(), tokbuf.size() * 8) || !cqs_persist::read_all(fd, s->off.data(), b_off) ||
        !cqs_persist::read_all(fd, post.data(), b_post) || !cqs_persist::read_all(fd, rankbuf.data(), rankbuf.size() * 8))
        return CQS_HIP_ERR_INVALID;
    ck.update(tokbuf.data(), tokbuf.size() * 8, false);
    ck.update(s->off.data(), b_off, false);
    ck.update(post.data(), b_post, false);
    ck.update(rankbuf.data(), rankbuf.size() * 8, false);
    ck.update(nullptr, 0, true);
    if (ck.finish() != h.checksum) return CQS_HIP_ERR_INVALID;     // corrupt body (index.rs:1035-1049)
    // structure: what the kernels rely on - ascending distinct tokens, offsets that tile the postings, positions inside the
    // index and ascending inside every list, a permutation for the id order
    s->tok.resize((size_t)h.tokens);
    memcpy(s->tok.data(), tokbuf.data(), b_tok);
    for (size_t t = 1; t < s->tok.size(); ++t)
        if (s->tok[t] <= s->tok[t - 1]) return CQS_HIP_ERR_INVALID;
    if (s->off[0] != 0 || s->off[(size_t)h.tokens] != h.postings) return CQS_HIP_ERR_INVALID;
    for (size_t t = 0; t < (size_t)h.tokens; ++t) {
        if (s->off[t + 1] < s->off[t] || s->off[t + 1] > h.postings) return CQS_HIP_ERR_INVALID;
        for (uint64_t e = s->off[t]; e < s->off[t + 1]; ++e) {
            if (post[e].x >= h.chunks || (e > s->off[t] && post[e].x < post[e - 1].x) || post[e].y == kUnscored) return CQS_HIP_ERR_INVALID;
        }
    }
    if (s->ranked) {
        s->chunk_of_rank.resize((size_t)h.chunks);
        memcpy(s->chunk_of_rank.data(), rankbuf.data(), b_rank);
        std::vector<uint8_t> seen((size_t)h.chunks, 0);
        for (uint32_t c : s->chunk_of_rank) {
            if (c >= h.chunks || seen[c]) return CQS_HIP_ERR_INVALID;
            seen[c] = 1;
        }
    }
    const int32_t rc = finish_create(s, post, device);
    if (rc != CQS_HIP_OK) return rc;
    guard.p = nullptr;
    *out = s;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

void cqs_hip_sparse_index_destroy(cqs_hip_sparse_index* s) CQS_ABI_TRY {
    release(s);
} CQS_ABI_CATCH_VOID

uint64_t cqs_hip_sparse_index_len(const cqs_hip_sparse_index* s) CQS_ABI_TRY {
    return s ? s->n : 0;
} CQS_ABI_CATCH_VAL(0)

uint64_t cqs_hip_sparse_index_unique_tokens(const cqs_hip_sparse_index* s) CQS_ABI_TRY {
    return s ? (uint64_t)s->tok.size() : 0;
} CQS_ABI_CATCH_VAL(0)

uint64_t cqs_hip_sparse_index_postings(const cqs_hip_sparse_index* s) CQS_ABI_TRY {
    return s ? s->n_postings : 0;
} CQS_ABI_CATCH_VAL(0)

int32_t cqs_hip_sparse_index_poisoned(const cqs_hip_sparse_index* s) CQS_ABI_TRY {
    if (!s) return 0;
    return s->poisoned.load(std::memory_order_acquire) ? 1 : 0;
} CQS_ABI_CATCH_VAL(0)

size_t cqs_hip_sparse_index_last_error(const cqs_hip_sparse_index* s, char* buf, size_t cap) CQS_ABI_TRY {
    if (!s) return 0;
    std::lock_guard<std::mutex> g(const_cast<cqs_hip_sparse_index*>(s)->mu);
    if (buf && cap) {
        const size_t m = std::min(cap - 1, s->last_error.size());
        memcpy(buf, s->last_error.data(), m);
        buf[m] = 0;
    }
    return s->last_error.size();
} CQS_ABI_CATCH_VAL(0)

namespace {

// Shared by the two search entry points; the caller holds s->mu.  q_off [b + 1]: the terms of query q are
// q_tokens / q_weights [q_off[q], q_off[q + 1]).
int32_t search_locked(cqs_hip_sparse_index* s, const uint64_t* q_off, const uint32_t* q_tokens, const float* q_weights, uint32_t b,
                      uint32_t k, const uint32_t* keep_bitset, uint64_t* out_chunks, float* out_scores, uint32_t* out_counts) {
    for (uint32_t q = 0; q < b; ++q) out_counts[q] = 0;
    if (s->poisoned) return CQS_HIP_ERR_POISONED;
    if (k > cqs::kMaxK) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: k > CQS_HIP_MAX_K");
    if (b > kSparseMaxBatch) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: more than 64 queries in a batch");
    const uint64_t total_terms = q_off[b] - q_off[0];
    for (uint32_t q = 0; q < b; ++q) {
        if (q_off[q + 1] < q_off[q]) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: query offsets not ascending");
        if (q_off[q + 1] - q_off[q] > kMaxTerms) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: too many query terms");
    }
    if (total_terms && (!q_tokens || !q_weights)) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: null query");
    s->last_ms = 0.f;
    s->last_touched = 0;
    if (total_terms == 0 || s->n == 0 || k == 0) return CQS_HIP_OK;    // index.rs:237-239; BoundedScoreHeap::new(0) keeps nothing
    if (!out_chunks || !out_scores) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: null output");
    S_TRY(s, hipSetDevice(s->device));
    int32_t rc = ensure_terms(s, (uint32_t)total_terms);
    if (rc != CQS_HIP_OK) return rc;
    if ((rc = ensure_batch(s, b)) != CQS_HIP_OK) return rc;
    // resolve the terms (`self.postings.get(&token_id)`, index.rs:249): a token without a list scores nothing
    uint32_t nt = 0;
    uint64_t touched = 0;
    for (uint32_t q = 0; q < b; ++q) {
        s->h_qoff[q] = nt;
        for (uint64_t i = q_off[q]; i < q_off[q + 1]; ++i) {
            uint32_t bits;
            memcpy(&bits, &q_weights[i], 4);
            if (bits == kUnscored) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: reserved NaN payload in a query weight");
            const auto it = std::lower_bound(s->tok.begin(), s->tok.end(), q_tokens[i]);
            if (it == s->tok.end() || *it != q_tokens[i]) continue;
            const size_t slot = (size_t)(it - s->tok.begin());
            const uint64_t len = s->off[slot + 1] - s->off[slot];
            if (len == 0) continue;
            if (len > 0xFFFFFFFFull) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: posting list longer than 2^32");
            s->h_terms[nt].start = s->off[slot];
            s->h_terms[nt].dir = s->dir_off[slot];
            s->h_terms[nt].len = (uint32_t)len;
            s->h_terms[nt].w = q_weights[i];
            ++nt;
            touched += len;
        }
    }
    s->h_qoff[b] = nt;
    if (nt == 0) return CQS_HIP_OK;
    s->last_touched = touched;
    hipStream_t st = s->stream;
    S_TRY(s, hipMemcpyAsync(s->d_qoff, s->h_qoff, kQoffBytes + (size_t)nt * sizeof(SparseTerm), hipMemcpyHostToDevice, st));
    const uint32_t* d_keep = nullptr;
    if (keep_bitset) {
        const size_t words = (size_t)((s->n + 31) / 32);
        memcpy(s->h_keep, keep_bitset, words * 4);
        S_TRY(s, hipMemcpyAsync(s->d_keep, s->h_keep, words * 4, hipMemcpyHostToDevice, st));
        d_keep = s->d_keep;
    }
    const uint32_t waves = s->n_pad / s->rw;
    // the scoring launch is bracketed by events only once somebody has asked for its time (cqs_hip_sparse_index_last_search
    // with a non-NULL accumulate_ms): two event records are ~4 us of an 80 us call
    const bool timed = s->want_timing.load(std::memory_order_relaxed);
    if (timed) S_TRY(s, hipEventRecord(s->ev0, st));
    hipLaunchKernelGGL(sparse_accumulate_kernel, dim3((waves + 3u) / 4u, b), dim3(256), (size_t)4 * sparse_wave_lds_words(s->rw) * 4, st,
                       s->d_post, s->d_terms, s->d_qoff, s->d_dir, (uint32_t)s->n, s->n_pad, s->sh, d_keep,
                       s->ranked ? s->d_chunk_of_rank : nullptr, s->d_scores, s->d_gmax,
                       (uint64_t*)(s->d_gmax + (size_t)s->b_cap * (s->n_pad / (s->group16 ? 16u : 64u))), s->group16 ? 1u : 0u);
    S_TRY(s, hipGetLastError());
    if (timed) S_TRY(s, hipEventRecord(s->ev1, st));
    cqs::ScanArgs a{};
    a.n = (uint32_t)s->n;
    a.n_pad = s->n_pad;
    a.b = b;
    a.scores = s->d_scores;
    a.gmax = s->d_gmax;
    a.gaux = (uint64_t*)(s->d_gmax + (size_t)s->b_cap * (s->n_pad / (s->group16 ? 16u : 64u)));
    a.gemv_only = true;                                   // (not the matrix-core scan: the select reads gaux)
    a.tiers = s->group16 ? cqs::TaskTiers{0u, 0u, s->n_pad / 16u} : cqs::TaskTiers{s->n_pad / 64u, 0u, 0u};
    a.k = k;
    a.linear_bins = false;
    a.range_bins = true;
    a.work = s->d_work;
    a.n_cu = s->n_cu;
    a.dbg = s->d_dbg;
    uint32_t* const h_counts = (uint32_t*)(s->h_out_keys + (size_t)s->b_cap * cqs::kMaxK);     // after the key block
    if (s->h_out_keys_dev) {
        S_TRY(s, cqs::launch_select(a, 0u, s->h_out_keys_dev, (uint32_t*)(s->h_out_keys_dev + (size_t)s->b_cap * cqs::kMaxK), st));
    } else {
        S_TRY(s, cqs::launch_select(a, 0u, s->d_out_keys, s->d_out_count, st));
        S_TRY(s, hipMemcpyAsync(s->h_out_keys, s->d_out_keys, (size_t)b * k * 8, hipMemcpyDeviceToHost, st));
        S_TRY(s, hipMemcpyAsync(h_counts, s->d_out_count, (size_t)b * 4, hipMemcpyDeviceToHost, st));
    }
    S_TRY(s, hipStreamSynchronize(st));
    if (timed) (void)hipEventElapsedTime(&s->last_ms, s->ev0, s->ev1);
    if (s->d_dbg) {
        unsigned long long h[16];
        if (hipMemcpy(h, s->d_dbg, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[cqs_hip sparse] select phases (us): zero+range %.2f | histogram %.2f | decide %.2f | groups %.2f | gather %.2f | sort %.2f | out %.2f | groups=%llu candidates=%llu\n",
                    0.0, (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01, (h[3] - h[2]) * 0.01, (h[4] - h[3]) * 0.01, (h[5] - h[4]) * 0.01,
                    (h[6] - h[5]) * 0.01, h[8], h[9]);
    }
    for (uint32_t q = 0; q < b; ++q) {                     // the select packs query q's keys at [q * k, (q + 1) * k)
        uint32_t cnt = h_counts[q];
        if (cnt > k) cnt = k;
        uint64_t* const oc = out_chunks + (size_t)q * k;
        cqs_hip_unpack_keys(s->h_out_keys + (size_t)q * k, cnt, oc, out_scores + (size_t)q * k);
        if (s->ranked)
            for (uint32_t i = 0; i < cnt; ++i) oc[i] = s->chunk_of_rank[(size_t)oc[i]];
        out_counts[q] = cnt;
    }
    return CQS_HIP_OK;
}

}  // namespace

namespace {

// ---- the combining queue: the dense index's scheme (index.hip) over the batched launches -----------------
// The reference's daemon calls `search_with_filter(&self)` from one thread per client on a shared index
// (src/cli/batch/view.rs:1621, src/search/query.rs:898-901).  An unfiltered single-query call parks its request; whoever
// leads next takes the device mutex first, gathers the parked requests with the same k (oldest first, up to 64) and runs
// them as one batch - every caller gets exactly the bits its own call would have produced (the kernels treat the queries
// of a batch independently).  Requests are validated BEFORE they park, so a batch can only fail for the device.
uint32_t count_like_front(const cqs_hip_sparse_index* s) {
    uint32_t n = 0;
    for (const cqs_sparse_req* r : s->pending) n += r->k == s->pending.front()->k ? 1u : 0u;
    return n;
}

// Lead one batch.  `lk` holds cmu on entry and on exit; s->leader is set by the caller.
void sparse_combine_lead(cqs_hip_sparse_index* s, std::unique_lock<std::mutex>& lk) {
    // stragglers of the last pass are on their way back: wait for them until combine_wait_us after that pass ENDED (a caller
    // that comes alone later than that does not wait), without the device mutex (round 5, as index.hip's combine_lead)
    const uint32_t target = s->expect < kSparseMaxBatch ? s->expect : kSparseMaxBatch;
    if (s->combine_wait_us && count_like_front(s) < target) {
        const auto t_end = s->last_pass_end + std::chrono::microseconds(s->combine_wait_us);
        while (count_like_front(s) < target && std::chrono::steady_clock::now() < t_end) {
            lk.unlock();
            for (int i = 0; i < 64; ++i) __builtin_ia32_pause();
            lk.lock();
        }
    }
    cqs_sparse_req* batch[kSparseMaxBatch];
    uint32_t nb = 0, left_like = 0;
    {
        const uint32_t k0 = s->pending.front()->k;
        std::deque<cqs_sparse_req*> keep;
        for (cqs_sparse_req* r : s->pending) {
            if (r->k == k0) {
                if (nb < kSparseMaxBatch) { batch[nb++] = r; continue; }
                ++left_like;
            }
            keep.push_back(r);
        }
        s->pending.swap(keep);
    }
    s->expect = nb + left_like;
    lk.unlock();
    std::unique_lock<std::mutex> dev(s->mu);               // the device, for the batch alone

    int32_t rc = CQS_HIP_OK;
    const uint32_t k = batch[0]->k;
    std::vector<uint32_t> counts(nb, 0u);
    std::vector<uint64_t> chunks;
    std::vector<float> scores;
    try {
        std::vector<uint64_t> q_off(nb + 1, 0);
        for (uint32_t i = 0; i < nb; ++i) q_off[i + 1] = q_off[i] + batch[i]->n_terms;
        std::vector<uint32_t> toks((size_t)q_off[nb]);
        std::vector<float> wts((size_t)q_off[nb]);
        for (uint32_t i = 0; i < nb; ++i) {
            if (batch[i]->n_terms == 0) continue;
            memcpy(toks.data() + q_off[i], batch[i]->q_tokens, (size_t)batch[i]->n_terms * 4);
            memcpy(wts.data() + q_off[i], batch[i]->q_weights, (size_t)batch[i]->n_terms * 4);
        }
        chunks.resize((size_t)nb * k);
        scores.resize((size_t)nb * k);
        rc = search_locked(s, q_off.data(), toks.data(), wts.data(), nb, k, nullptr, chunks.data(), scores.data(), counts.data());
    } catch (const std::bad_alloc&) {
        rc = sfail(s, CQS_HIP_ERR_NOMEM, "sparse search: out of host memory");
    } catch (...) {
        rc = sfail(s, CQS_HIP_ERR_INVALID, "sparse search: unexpected C++ exception");
    }
    s->stat_passes.fetch_add(1, std::memory_order_relaxed);
    s->stat_queries.fetch_add(nb, std::memory_order_relaxed);
    const bool poisoned = s->poisoned.load(std::memory_order_acquire);
    if (rc == CQS_HIP_OK)
        for (uint32_t i = 0; i < nb; ++i) {
            memcpy(batch[i]->out_chunks, chunks.data() + (size_t)i * k, (size_t)counts[i] * 8);
            memcpy(batch[i]->out_scores, scores.data() + (size_t)i * k, (size_t)counts[i] * 4);
            *batch[i]->out_count = counts[i];
        }
    dev.unlock();

    lk.lock();
    s->last_pass_end = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < nb; ++i) {
        batch[i]->rc = (rc != CQS_HIP_OK && i > 0 && poisoned) ? CQS_HIP_ERR_POISONED : rc;
        batch[i]->done = true;
    }
    if (poisoned) {                                        // nobody stays parked on a dead handle
        for (cqs_sparse_req* r : s->pending) { r->rc = CQS_HIP_ERR_POISONED; r->done = true; }
        s->pending.clear();
    } else if (!s->pending.empty()) {
        uint32_t like = 0;
        for (const cqs_sparse_req* r : s->pending) like += r->k == k ? 1u : 0u;
        if (nb + like > s->expect) s->expect = nb + like;
    }
}

int32_t sparse_combine_search(cqs_hip_sparse_index* s, cqs_sparse_req& r) {
    std::unique_lock<std::mutex> lk(s->cmu);
    s->pending.push_back(&r);
    while (!r.done) {
        if (!s->leader) {
            s->leader = true;
            struct Reset {                                 // whatever happens in there, the next caller can lead
                cqs_hip_sparse_index* s; std::unique_lock<std::mutex>& lk;
                ~Reset() { if (!lk.owns_lock()) lk.lock(); s->leader = false; s->ccv.notify_all(); }
            } reset{s, lk};
            sparse_combine_lead(s, lk);
        } else {
            s->ccv.wait(lk);
        }
    }
    return r.rc;
}

}  // namespace

int32_t cqs_hip_sparse_index_search(cqs_hip_sparse_index* s, const uint32_t* q_tokens, const float* q_weights, uint32_t n_terms,
                                    uint32_t k, const uint32_t* keep_bitset, uint64_t* out_chunks, float* out_scores,
                                    uint32_t* out_count) CQS_ABI_TRY {
    if (!s) return CQS_HIP_ERR_INVALID;
    // One unfiltered query with its arguments in order: the combining queue.  Everything a batch could refuse for ONE of
    // its members is checked here, before the request parks.
    if (s->combine && !keep_bitset && out_count && out_chunks && out_scores && k >= 1 && k <= cqs::kMaxK && n_terms >= 1 &&
        n_terms <= kMaxTerms && q_tokens && q_weights && s->n != 0) {
        if (s->poisoned.load(std::memory_order_acquire)) { *out_count = 0; return CQS_HIP_ERR_POISONED; }
        bool reserved = false;
        for (uint32_t i = 0; i < n_terms; ++i) {
            uint32_t bits;
            memcpy(&bits, &q_weights[i], 4);
            reserved |= bits == kUnscored;
        }
        if (!reserved) {
            *out_count = 0;
            cqs_sparse_req r{q_tokens, q_weights, n_terms, k, out_chunks, out_scores, out_count};
            return sparse_combine_search(s, r);
        }
    }
    std::lock_guard<std::mutex> g(s->mu);
    if (!out_count) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: null out_count");
    const uint64_t q_off[2] = {0, n_terms};
    return search_locked(s, q_off, q_tokens, q_weights, 1u, k, keep_bitset, out_chunks, out_scores, out_count);
} CQS_ABI_CATCH(s)

int32_t cqs_hip_sparse_index_search_batch(cqs_hip_sparse_index* s, const uint64_t* q_off, const uint32_t* q_tokens,
                                          const float* q_weights, uint32_t b, uint32_t k, const uint32_t* keep_bitset,
                                          uint64_t* out_chunks, float* out_scores, uint32_t* out_counts) CQS_ABI_TRY {
    if (!s) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->mu);
    if (b == 0) return CQS_HIP_OK;
    if (!q_off || !out_counts) return sfail(s, CQS_HIP_ERR_INVALID, "sparse search: null offsets / counts");
    return search_locked(s, q_off, q_tokens, q_weights, b, k, keep_bitset, out_chunks, out_scores, out_counts);
} CQS_ABI_CATCH(s)

void cqs_hip_sparse_index_combine_stats(const cqs_hip_sparse_index* s, uint64_t* passes, uint64_t* queries) CQS_ABI_TRY {
    if (passes) *passes = s ? s->stat_passes.load(std::memory_order_relaxed) : 0;
    if (queries) *queries = s ? s->stat_queries.load(std::memory_order_relaxed) : 0;
} CQS_ABI_CATCH_VOID

int32_t cqs_hip_sparse_index_last_search(const cqs_hip_sparse_index* s, float* accumulate_ms, uint64_t* touched_postings) CQS_ABI_TRY {
    if (!s) return CQS_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> g(const_cast<cqs_hip_sparse_index*>(s)->mu);
    if (accumulate_ms) {
        *accumulate_ms = s->last_ms;             // 0 for a search that ran before the first request
        const_cast<cqs_hip_sparse_index*>(s)->want_timing.store(true, std::memory_order_relaxed);
    }
    if (touched_postings) *touched_postings = s->last_touched;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_VAL(CQS_HIP_ERR_INVALID)

// Diagnostic (not in the header): n_threads native threads, each calling the public single-query entry point per_thread
// times over its own residue class of the query set (query q's terms: [q_off[q], q_off[q + 1])); results land in the
// query's own output slot.  Returns the wall time in seconds, < 0 on a failed call.  What bench.py's Python threads cannot
// show behind the interpreter lock.
double cqs_hip_debug_sparse_client_storm(cqs_hip_sparse_index* s, const uint64_t* q_off, const uint32_t* q_tokens, const float* q_weights,
                                         uint32_t n_queries, uint32_t k, uint32_t n_threads, uint32_t per_thread,
                                         uint64_t* out_chunks, float* out_scores, uint32_t* out_counts) CQS_ABI_TRY {
    if (!s || !q_off || !q_tokens || !q_weights || !n_queries || !n_threads || !out_chunks || !out_scores || !out_counts) return -1.0;
    std::atomic<int32_t> bad{0};
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> th;
    th.reserve(n_threads);
    for (uint32_t t = 0; t < n_threads; ++t)
        th.emplace_back([&, t]() {
            ready.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            uint32_t qi = t % n_queries;
            for (uint32_t i = 0; i < per_thread; ++i) {
                const int32_t rc = cqs_hip_sparse_index_search(s, q_tokens + q_off[qi], q_weights + q_off[qi], (uint32_t)(q_off[qi + 1] - q_off[qi]),
                                                               k, nullptr, out_chunks + (size_t)qi * k, out_scores + (size_t)qi * k, out_counts + qi);
                if (rc != CQS_HIP_OK) { bad.store(rc); break; }
                qi = (qi + n_threads) % n_queries;
            }
        });
    while (ready.load() < n_threads) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (std::thread& t : th) t.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return bad.load() ? -1.0 : el;
} CQS_ABI_CATCH_VAL(-1.0)

}  // extern "C"
