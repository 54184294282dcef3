// scan_kernels.hip — gfx950 (MI355X) kernels for the cqs brute-force scan.
//
//  scan_gemv_kernel   HBM-streaming fp32 dot of every corpus row with 1..8
//                     queries.  One wave owns 64 consecutive rows; a row is
//                     read as dim/256 fully coalesced 1-KiB wave loads
//                     (16 B/lane), the query lives in registers, lane partials
//                     are reduced with a transposed butterfly so that 4 rows x
//                     BQ queries cost ~1 cross-lane op per dot.  Replaces the
//                     per-row simsimd dot of the reference's brute-force loop
//                     (src/math.rs:11-28 called from src/search/query.rs:469-481)
//                     and cuVS' search for the exact backend (src/cagra.rs:605).
//  select_*           exact top-k over the score rows: 12-bit radix histograms
//                     (threshold search), candidate compaction, bitonic sort.
//                     Replaces BoundedScoreHeap (candidate.rs:162-330): same
//                     comparator (score desc under total order, id asc) on a
//                     packed 64-bit key.
//
// Wave = 64 lanes.  No CUDA-isms, no dual paths: gfx950 only.
#include "scan_kernels.h"

namespace cqs {

typedef float f4 __attribute__((ext_vector_type(4)));

// ---- ordered keys ----------------------------------------------------------
// f32 -> u32 preserving IEEE total order (what Rust's f32::total_cmp sorts by).
__device__ __forceinline__ uint32_t okey(float x) {
    uint32_t b = __float_as_uint(x);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
// Dropped entries are stored as -inf: okey(-inf) = 0x007FFFFF.  Every finite
// score has a larger key; +inf / NaN never reach the score rows.
constexpr uint32_t kInvalidKey = 0x007FFFFFu;

__device__ __forceinline__ uint64_t pack_key(uint32_t ok, uint32_t global_row) {
    return ((uint64_t)ok << 32) | (uint64_t)(0xFFFFFFFFu - global_row);
}

// ---- transposed butterfly reduction ---------------------------------------
// v[0..NV) hold per-lane partial sums of NV independent dot products.  After
// the call v[0] of lane L is the complete sum of product number L / (64/NV).
// Cost: NV-1 + log2(64/NV) cross-lane ops instead of 6*NV.
template <int NV>
__device__ __forceinline__ void treduce(float (&v)[NV], int lane) {
    int m = 32;
#pragma unroll
    for (int n = NV; n > 1; n >>= 1) {
        const bool hi = (lane & m) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float keep = hi ? v[i + n / 2] : v[i];
            const float send = hi ? v[i] : v[i + n / 2];
            v[i] = keep + __shfl_xor(send, m, 64);
        }
        m >>= 1;
    }
#pragma unroll
    for (; m >= 1; m >>= 1) v[0] += __shfl_xor(v[0], m, 64);
}

// ---- scan ------------------------------------------------------------------
// NCH = ceil(dim / 256): 1-KiB chunks per row.  BQ queries, RI rows per inner
// iteration (RI*BQ partial sums are reduced together).
template <int NCH, int BQ, int RI, bool NT>
__global__ __launch_bounds__(256) void scan_gemv_kernel(
    const float* __restrict__ rows, uint32_t n, uint32_t n_pad, uint32_t dim,
    const float* __restrict__ q, float* __restrict__ scores,
    const uint32_t* __restrict__ keep, uint32_t mode, float thr) {
    constexpr int NV = RI * BQ;
    constexpr int LPV = 64 / NV;  // lanes per reduced value
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const uint32_t base = (blockIdx.x * 4u + (uint32_t)wid) * 64u;  // first row of this wave
    if (base >= n_pad) return;

    const bool full = (dim == (uint32_t)NCH * 256u);  // no partial last chunk
    // query fragments: lane owns floats [c*256 + lane*4, +4) of every chunk c
    f4 qv[BQ][NCH];
#pragma unroll
    for (int b = 0; b < BQ; ++b)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t idx = (uint32_t)c * 256u + (uint32_t)lane * 4u;
            qv[b][c] = (full || idx < dim) ? *(const f4*)(q + (size_t)b * dim + idx) : (f4)(0.f);
        }

    // rows this wave must score: inside the corpus and kept by the filter
    uint64_t mask = ~0ull;
    if (base + 64u > n) mask = (base >= n) ? 0ull : (~0ull >> (64u - (n - base)));
    if (keep) {
        const uint32_t nwords = (n + 31u) / 32u;
        const uint32_t w = base / 32u;
        const uint32_t w0 = (w < nwords) ? keep[w] : 0u;
        const uint32_t w1 = (w + 1u < nwords) ? keep[w + 1u] : 0u;
        mask &= ((uint64_t)w1 << 32) | (uint64_t)w0;
    }
    // wave-uniform by construction; tell the compiler so branches are scalar
    const uint32_t mlo = __builtin_amdgcn_readfirstlane((uint32_t)mask);
    const uint32_t mhi = __builtin_amdgcn_readfirstlane((uint32_t)(mask >> 32));
    mask = ((uint64_t)mhi << 32) | mlo;

    float sc[BQ];
#pragma unroll
    for (int b = 0; b < BQ; ++b) sc[b] = -INFINITY;

    const uint32_t last = n - 1u;
    for (int j = 0; j < 64 / RI; ++j) {
        const uint32_t m = (uint32_t)(mask >> (RI * j)) & ((1u << RI) - 1u);
        if (m == 0u) continue;  // all RI rows filtered out / past the end: skip their HBM reads
        float acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = 0.f;
        f4 x[RI][NCH];
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            uint32_t row = base + (uint32_t)(RI * j + r);
            row = row > last ? last : row;
            const float* p = rows + (size_t)row * dim;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t idx = (uint32_t)c * 256u + (uint32_t)lane * 4u;
                if (full || idx < dim) {
                    if (NT) x[r][c] = __builtin_nontemporal_load((const f4*)(p + idx));
                    else x[r][c] = *(const f4*)(p + idx);
                } else {
                    x[r][c] = (f4)(0.f);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int b = 0; b < BQ; ++b) {
                    float a = acc[b * RI + r];
                    a = __builtin_fmaf(x[r][c].x, qv[b][c].x, a);
                    a = __builtin_fmaf(x[r][c].y, qv[b][c].y, a);
                    a = __builtin_fmaf(x[r][c].z, qv[b][c].z, a);
                    a = __builtin_fmaf(x[r][c].w, qv[b][c].w, a);
                    acc[b * RI + r] = a;
                }
        treduce<NV>(acc, lane);
        // value (b, r) now sits in lanes [(b*RI+r)*LPV, +LPV); lane L = RI*j + r wants it
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            const float t = __shfl(acc[0], (b * RI + (lane % RI)) * LPV, 64);
            if (lane / RI == j) sc[b] = t;
        }
    }

    // epilogue: lane <-> row base+lane; one coalesced 256-B store per query
    const uint32_t row = base + (uint32_t)lane;
    const bool live = (mask >> lane) & 1ull;
#pragma unroll
    for (int b = 0; b < BQ; ++b) {
        float s = sc[b];
        // non-finite scores are never emitted (src/math.rs:23-27, src/cagra.rs:649-651)
        if (!live || !(__builtin_fabsf(s) <= 3.4028234664e38f)) s = -INFINITY;
        else if (mode == 1u) {
            // candidate.rs:550 clamp(0,1) (Rust clamp keeps -0.0), :513-519 `>= threshold`
            s = s < 0.f ? 0.f : (s > 1.f ? 1.f : s);
            if (!(s >= thr)) s = -INFINITY;
        }
        scores[(size_t)b * n_pad + row] = s;
    }
}

// ---- select: block-wide "find the bin holding the k-th largest" -------------
// hist: kHistBins counters (global or LDS).  Finds T = the highest bin such
// that count(bins >= T) >= k_rem.  res[0]=T res[1]=count(bins > T) res[2]=hist[T]
// res[3]=total count.  If total < k_rem: T = 0, res[1] = total - hist[0].
template <int THREADS>
__device__ void block_decide(const uint32_t* hist, uint32_t k_rem, uint32_t* s_part /*THREADS*/,
                             uint32_t* res /*4, LDS*/) {
    constexpr int BPT = kHistBins / THREADS;
    const int t = threadIdx.x;
    uint32_t h[BPT];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        h[i] = hist[t * BPT + i];
        sum += h[i];
    }
    // inclusive suffix scan over threads (thread THREADS-1 owns the top bins)
    s_part[t] = sum;
    __syncthreads();
    for (int off = 1; off < THREADS; off <<= 1) {
        uint32_t add = (t + off < THREADS) ? s_part[t + off] : 0u;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    const uint32_t incl = s_part[t];
    uint32_t above = incl - sum;  // count in bins owned by higher threads
    if (t == 0) {
        res[3] = incl;
        if (incl < k_rem) {  // fewer valid entries than requested: take them all
            res[0] = 0;
            res[1] = incl - h[0];
            res[2] = h[0];
        }
    }
    if (above < k_rem && incl >= k_rem) {  // exactly one thread
#pragma unroll
        for (int i = BPT - 1; i >= 0; --i) {
            if (above < k_rem && above + h[i] >= k_rem) {
                res[0] = (uint32_t)(t * BPT + i);
                res[1] = above;
                res[2] = h[i];
            }
            above += h[i];
        }
    }
    __syncthreads();
}

// Decision shared (recomputed) by hist2 / collect: lower bound LB on the
// ordered score key such that {key >= LB} holds the top k and, when `ok`,
// at most kCandCap entries.
struct SelPlan {
    uint32_t lb;       // candidates: okey >= lb (and valid)
    uint32_t need_l2;  // level-1 bin too crowded: level 2 refines inside bin t1
    uint32_t t1;
    uint32_t k_rem;    // k minus entries above bin t1
    uint32_t above1;
};

template <int THREADS>
__device__ SelPlan plan_level1(const uint32_t* hist1, uint32_t k, uint32_t* s_part, uint32_t* res) {
    block_decide<THREADS>(hist1, k, s_part, res);
    SelPlan p;
    p.t1 = res[0];
    p.above1 = res[1];
    const uint32_t cnt = res[2], total = res[3];
    __syncthreads();
    p.k_rem = k - (p.above1 < k ? p.above1 : k);
    if (total <= k) {  // everything valid is selected
        p.lb = 0;
        p.need_l2 = 0;
    } else {
        p.lb = p.t1 << 20;
        p.need_l2 = (p.above1 + cnt > kCandCap) ? 1u : 0u;
    }
    return p;
}

// level-1 histogram of the top 12 key bits of every valid score
__global__ __launch_bounds__(256) void select_hist1_kernel(const float* __restrict__ scores, uint32_t n_pad,
                                                           uint32_t seg, uint32_t* __restrict__ sel) {
    __shared__ uint32_t s_hist[kHistBins];
    const uint32_t qi = blockIdx.y;
    for (int i = threadIdx.x; i < (int)kHistBins; i += 256) s_hist[i] = 0;
    __syncthreads();
    const float* s = scores + (size_t)qi * n_pad;
    const uint32_t lo = blockIdx.x * seg;
    uint32_t hi = lo + seg;
    if (hi > n_pad) hi = n_pad;
    for (uint32_t i = lo + threadIdx.x * 4u; i < hi; i += 1024u) {
        const f4 v = *(const f4*)(s + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t kk = okey(v[e]);
            if (kk > kInvalidKey) atomicAdd(&s_hist[kk >> 20], 1u);
        }
    }
    __syncthreads();
    uint32_t* g = sel + (size_t)qi * kSelWords;
    for (int i = threadIdx.x; i < (int)kHistBins; i += 256) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&g[i], c);
    }
}

// level-2 histogram (key bits 19..8) inside level-1 bin t1; no-op unless needed
__global__ __launch_bounds__(256) void select_hist2_kernel(const float* __restrict__ scores, uint32_t n_pad,
                                                           uint32_t seg, uint32_t k, uint32_t* __restrict__ sel) {
    __shared__ uint32_t s_hist[kHistBins];
    __shared__ uint32_t s_part[256];
    __shared__ uint32_t s_res[4];
    const uint32_t qi = blockIdx.y;
    uint32_t* g = sel + (size_t)qi * kSelWords;
    const SelPlan p = plan_level1<256>(g, k, s_part, s_res);
    if (!p.need_l2) return;
    for (int i = threadIdx.x; i < (int)kHistBins; i += 256) s_hist[i] = 0;
    __syncthreads();
    const float* s = scores + (size_t)qi * n_pad;
    const uint32_t lo = blockIdx.x * seg;
    uint32_t hi = lo + seg;
    if (hi > n_pad) hi = n_pad;
    for (uint32_t i = lo + threadIdx.x * 4u; i < hi; i += 1024u) {
        const f4 v = *(const f4*)(s + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t kk = okey(v[e]);
            if (kk > kInvalidKey && (kk >> 20) == p.t1) atomicAdd(&s_hist[(kk >> 8) & 0xFFFu], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (int)kHistBins; i += 256) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&g[kHistBins + i], c);
    }
}

// compaction of every entry with key >= LB into the query's candidate list
__global__ __launch_bounds__(256) void select_collect_kernel(const float* __restrict__ scores, uint32_t n_pad,
                                                             uint32_t seg, uint32_t k, uint32_t row_base,
                                                             uint32_t* __restrict__ sel,
                                                             uint64_t* __restrict__ cand) {
    __shared__ uint32_t s_part[256];
    __shared__ uint32_t s_res[4];
    const uint32_t qi = blockIdx.y;
    uint32_t* g = sel + (size_t)qi * kSelWords;
    SelPlan p = plan_level1<256>(g, k, s_part, s_res);
    uint32_t lb = p.lb;
    if (p.need_l2) {
        block_decide<256>(g + kHistBins, p.k_rem, s_part, s_res);
        lb = (p.t1 << 20) | (s_res[0] << 8);
        __syncthreads();
    }
    uint32_t* cnt = g + 2 * kHistBins;
    uint64_t* out = cand + (size_t)qi * kCandCap;
    const float* s = scores + (size_t)qi * n_pad;
    const uint32_t lo = blockIdx.x * seg;
    uint32_t hi = lo + seg;
    if (hi > n_pad) hi = n_pad;
    const int lane = threadIdx.x & 63;
    for (uint32_t i0 = lo; i0 < hi; i0 += 1024u) {  // uniform trip count per block
        const uint32_t i = i0 + threadIdx.x * 4u;
        f4 v = (f4)(-INFINITY);
        if (i < hi) v = *(const f4*)(s + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t kk = okey(v[e]);
            const bool take = (kk > kInvalidKey) && (kk >= lb);
            const unsigned long long bal = __ballot(take);
            if (bal) {
                const uint32_t tot = (uint32_t)__popcll(bal);
                uint32_t off = 0;
                if (lane == 0) off = atomicAdd(cnt, tot);
                off = __shfl(off, 0, 64);
                const uint32_t rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                if (take && off + rank < kCandCap) out[off + rank] = pack_key(kk, row_base + i + (uint32_t)e);
            }
        }
    }
}

// ---- one-block exact select (fallback for heavy ties / crowded bins) --------
// Radix select on the full 64-bit packed key (score bits then row bits: all
// keys distinct), 12-bit digits, streaming the whole score row per pass.
// Leaves <= kCandCap candidates in s_keys and returns their count.
__device__ uint32_t slow_select(const float* __restrict__ s, uint32_t n_pad, uint32_t k, uint32_t row_base,
                                uint64_t* s_keys, uint32_t* s_hist, uint32_t* s_part, uint32_t* s_res,
                                uint32_t* s_cnt) {
    uint64_t prefix = 0;  // digits decided so far (top bits of the key)
    int bits_done = 0;
    uint32_t k_rem = k, sel_above = 0;
    uint64_t lb = 0;
    while (bits_done < 64) {
        const int w = (64 - bits_done) >= 12 ? 12 : (64 - bits_done);
        for (int i = threadIdx.x; i < (int)kHistBins; i += 1024) s_hist[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_pad; i += 1024u) {
            const uint32_t kk = okey(s[i]);
            if (kk <= kInvalidKey) continue;
            const uint64_t key = pack_key(kk, row_base + i);
            if (bits_done == 0 || (key >> (64 - bits_done)) == prefix)
                atomicAdd(&s_hist[(uint32_t)(key >> (64 - bits_done - w)) & ((1u << w) - 1u)], 1u);
        }
        __syncthreads();
        block_decide<1024>(s_hist, k_rem, s_part, s_res);
        const uint32_t T = s_res[0], above = s_res[1], cnt = s_res[2], total = s_res[3];
        __syncthreads();
        if (bits_done == 0 && total <= k) { lb = 0; break; }
        prefix = (prefix << w) | T;
        bits_done += w;
        sel_above += above;
        k_rem -= above;
        lb = prefix << (64 - bits_done);
        if (sel_above + cnt <= kCandCap) break;
    }
    if (threadIdx.x == 0) *s_cnt = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_pad; i += 1024u) {
        const uint32_t kk = okey(s[i]);
        if (kk <= kInvalidKey) continue;
        const uint64_t key = pack_key(kk, row_base + i);
        if (key >= lb) {
            const uint32_t slot = atomicAdd(s_cnt, 1u);
            if (slot < kCandCap) s_keys[slot] = key;
        }
    }
    __syncthreads();
    const uint32_t c = *s_cnt;
    return c < kCandCap ? c : kCandCap;
}

// final: sort the candidates (bitonic, descending) and emit the top k
__global__ __launch_bounds__(1024) void select_sort_kernel(const float* __restrict__ scores, uint32_t n_pad,
                                                           uint32_t k, uint32_t row_base,
                                                           const uint32_t* __restrict__ sel,
                                                           const uint64_t* __restrict__ cand,
                                                           uint64_t* __restrict__ out_keys,
                                                           uint32_t* __restrict__ out_counts) {
    __shared__ uint64_t s_keys[kCandCap];
    __shared__ uint32_t s_hist[kHistBins];
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_res[4];
    __shared__ uint32_t s_cnt;
    const uint32_t qi = blockIdx.x;
    const uint32_t raw = sel[(size_t)qi * kSelWords + 2 * kHistBins];
    uint32_t count;
    if (raw > kCandCap) {
        count = slow_select(scores + (size_t)qi * n_pad, n_pad, k, row_base, s_keys, s_hist, s_part, s_res, &s_cnt);
    } else {
        count = raw;
        const uint64_t* c = cand + (size_t)qi * kCandCap;
        for (uint32_t i = threadIdx.x; i < count; i += 1024u) s_keys[i] = c[i];
    }
    uint32_t P = 64;
    while (P < count) P <<= 1;
    for (uint32_t i = count + threadIdx.x; i < P; i += 1024u) s_keys[i] = 0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < P / 2; t += 1024u) {
                const uint32_t i = 2u * t - (t & (stride - 1u));  // lower index of the pair
                const uint32_t j = i + stride;
                const uint64_t a = s_keys[i], b = s_keys[j];
                const bool desc = ((i & size) == 0u);
                if ((a < b) == desc) {
                    s_keys[i] = b;
                    s_keys[j] = a;
                }
            }
            __syncthreads();
        }
    }
    const uint32_t outc = count < k ? count : k;
    for (uint32_t i = threadIdx.x; i < k; i += 1024u) out_keys[(size_t)qi * k + i] = (i < outc) ? s_keys[i] : 0ull;
    if (threadIdx.x == 0) out_counts[qi] = outc;
}

// ---- launchers -------------------------------------------------------------
bool scan_dim_supported(uint32_t dim) { return dim >= 4 && dim % 4 == 0 && dim <= 2048; }

template <int NCH, int BQ, int RI>
static hipError_t launch_gemv(const ScanArgs& a, const float* q, float* scores, hipStream_t st) {
    const dim3 grid(a.n_pad / kRowsPerBlock), block(256);
    if (a.nontemporal)
        hipLaunchKernelGGL((scan_gemv_kernel<NCH, BQ, RI, true>), grid, block, 0, st, a.rows, a.n, a.n_pad, a.dim, q,
                           scores, a.keep, a.mode, a.threshold);
    else
        hipLaunchKernelGGL((scan_gemv_kernel<NCH, BQ, RI, false>), grid, block, 0, st, a.rows, a.n, a.n_pad, a.dim, q,
                           scores, a.keep, a.mode, a.threshold);
    return hipGetLastError();
}

template <int NCH>
static hipError_t launch_gemv_groups(const ScanArgs& a, hipStream_t st) {
    uint32_t done = 0;
    while (done < a.b) {
        const uint32_t left = a.b - done;
        const float* q = a.q + (size_t)done * a.dim;
        float* sc = a.scores + (size_t)done * a.n_pad;
        hipError_t e;
        uint32_t g;
        // register budget ~ 4*NCH*(BQ + RI) + BQ*RI VGPRs: wide rows take fewer queries per pass
        if constexpr (NCH <= 4) {
            if (left >= 8) { g = 8; e = launch_gemv<NCH, 8, 2>(a, q, sc, st); }
            else if (left >= 4) { g = 4; e = launch_gemv<NCH, 4, 4>(a, q, sc, st); }
            else if (left >= 2) { g = 2; e = launch_gemv<NCH, 2, 4>(a, q, sc, st); }
            else { g = 1; e = launch_gemv<NCH, 1, 4>(a, q, sc, st); }
        } else {
            if (left >= 2) { g = 2; e = launch_gemv<NCH, 2, 2>(a, q, sc, st); }
            else { g = 1; e = launch_gemv<NCH, 1, 2>(a, q, sc, st); }
        }
        if (e != hipSuccess) return e;
        done += g;
    }
    return hipSuccess;
}

hipError_t launch_scan(const ScanArgs& a, hipStream_t st) {
    if (a.b == 0 || a.n == 0) return hipSuccess;
    const uint32_t nch = (a.dim + 255u) / 256u;
    switch (nch) {
        case 1: return launch_gemv_groups<1>(a, st);
        case 2: return launch_gemv_groups<2>(a, st);
        case 3: return launch_gemv_groups<3>(a, st);
        case 4: return launch_gemv_groups<4>(a, st);
        case 5: return launch_gemv_groups<5>(a, st);
        case 6: return launch_gemv_groups<6>(a, st);
        case 7: return launch_gemv_groups<7>(a, st);
        case 8: return launch_gemv_groups<8>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_select(const float* scores, uint32_t n, uint32_t n_pad, uint32_t b, uint32_t k,
                         uint32_t row_base, uint32_t* sel, uint64_t* cand, uint64_t* out_keys,
                         uint32_t* out_counts, hipStream_t st) {
    (void)n;
    if (b == 0 || k == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(sel, 0, (size_t)b * kSelWords * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    // segment per block: multiple of 1024 entries, at most 256 blocks per query
    uint32_t seg = (n_pad + 255u) / 256u;
    seg = ((seg + 1023u) / 1024u) * 1024u;
    if (seg < 4096u) seg = 4096u;
    const uint32_t nb = (n_pad + seg - 1u) / seg;
    const dim3 grid(nb, b), block(256);
    hipLaunchKernelGGL(select_hist1_kernel, grid, block, 0, st, scores, n_pad, seg, sel);
    hipLaunchKernelGGL(select_hist2_kernel, grid, block, 0, st, scores, n_pad, seg, k, sel);
    hipLaunchKernelGGL(select_collect_kernel, grid, block, 0, st, scores, n_pad, seg, k, row_base, sel, cand);
    hipLaunchKernelGGL(select_sort_kernel, dim3(b), dim3(1024), 0, st, scores, n_pad, k, row_base, sel, cand, out_keys,
                       out_counts);
    return hipGetLastError();
}

}  // namespace cqs
