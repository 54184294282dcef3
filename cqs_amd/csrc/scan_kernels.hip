// scan_kernels.hip — gfx950 (MI355X) kernels for the cqs brute-force scan.
//
//  scan_gemv_kernel   HBM-streaming fp32 dot of every corpus row with 1..8 queries.
//                     One wave = one task of 64 (32, 16) rows; up to ~1.5M rows every task gets
//                     its own wave and the hardware dispatcher does the scheduling, beyond that
//                     a persistent grid continues from a global work queue.  A row is read as
//                     dim/256 fully coalesced 1-KiB wave loads (16 B/lane) issued 8 rows at a
//                     time, double-buffered; the query lives in registers; lane partials are
//                     reduced with a transposed butterfly so that RI rows x BQ queries cost ~1
//                     cross-lane op per dot.  Besides the score row it emits one maximum per
//                     task: the pruning index of the top-k select.
//                     Replaces the per-row simsimd dot of the reference's brute-force loop
//                     (src/math.rs:11-28 called from src/search/query.rs:469-481) and
//                     cuVS' search for the exact backend (src/cagra.rs:605).
//  select_finish      exact top-k in ONE workgroup per query: threshold bin from the group-
//                     maxima histogram -> the few groups that can hold a top-k entry ->
//                     their scores -> bitonic sort (exact one-block radix fallback for ties).
//                     Replaces BoundedScoreHeap (candidate.rs:162-330): same comparator
//                     (score desc under total order, id asc) on a packed 64-bit key.
//
// Wave = 64 lanes.  gfx950 only.
#include "scan_kernels.h"
#include "launch_util.h"

namespace cqs {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) uint32_t gu32;

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

// Histogram bin of a valid score; monotone non-decreasing in the score.
// linear: 4096 bins of width 2^-11 over [-1,1] (cosine / clamped scores: fine
// resolution exactly where the top-k threshold lives); else the top 12 bits of
// the ordered key (log-spaced, any range: raw dot products).
__device__ __forceinline__ uint32_t bin_of(float s, bool linear) {
    if (linear) {
        const float t = (s + 1.0f) * 2048.0f;
        const int b = (int)t;  // t >= 0 for s >= -1; negatives truncate toward 0 and clamp below
        return (uint32_t)(b < 0 ? 0 : (b > 4095 ? 4095 : b));
    }
    return okey(s) >> 20;
}
// Third form (select mode 2, the sparse index): linear bins over the range the GROUP MAXIMA of this very row span - sums of
// SPLADE products crowd into two or three octaves, where the log-spaced bins above put thousands of groups into the
// threshold bin.  Any monotone map is correct (the candidates are re-ranked on their full keys); this one is sharp where
// the top-k lives.  Scores below `lo` fall into bin 0.
__device__ __forceinline__ uint32_t bin_of_range(float s, float lo, float scale) {
    if (!(scale > 0.f)) return 0u;                        // degenerate range: one bin
    const float t = (s - lo) * scale;
    if (!(t < 4095.0f)) return 4095u;                     // (also an overflowed difference)
    return t > 0.f ? (uint32_t)(int)t : 0u;
}
__device__ __forceinline__ uint32_t bin_any(float s, uint32_t mode, float lo, float scale) {
    return mode == 2u ? bin_of_range(s, lo, scale) : bin_of(s, mode != 0u);
}

// ---- transposed butterfly reduction ---------------------------------------
// v[0..NV) hold per-lane partial sums of NV independent dot products.  After
// the call v[0] of lane L is the complete sum of product number L / (64/NV).
// Cost: NV-1 + log2(64/NV) cross-lane ops instead of 6*NV.
template <int N, int M, int NV>
__device__ __forceinline__ void treduce_level(float (&v)[NV], int lane) {
    // Compile-time level (N live values, exchange distance M): a runtime loop over the levels makes
    // v[i + n/2] a variable index, which hipcc lowers to an NV-way v_cmp/v_cndmask chain per element.
    if constexpr (N > 1) {
        const bool hi = (lane & M) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const float keep = hi ? v[i + N / 2] : v[i];
            const float send = hi ? v[i] : v[i + N / 2];
            v[i] = keep + __shfl_xor(send, M, 64);
        }
        treduce_level<N / 2, M / 2, NV>(v, lane);
    } else if constexpr (M >= 1) {
        v[0] += __shfl_xor(v[0], M, 64);
        treduce_level<1, M / 2, NV>(v, lane);
    }
}
template <int NV>
__device__ __forceinline__ void treduce(float (&v)[NV], int lane) {
    treduce_level<NV, 32, NV>(v, lane);
}

// ---- block-wide "find the bin holding the k-th largest" ---------------------
// hist: kHistBins counters (global or LDS).  Finds T = the highest bin such
// that count(bins >= T) >= k_rem.  res[0]=T res[1]=count(bins > T) res[2]=hist[T]
// res[3]=total count.  If total < k_rem: T = 0, res[1] = total - hist[0].
template <int THREADS>
__device__ void block_decide(const uint32_t* hist, uint32_t k_rem, uint32_t* s_part /*>= THREADS/64*/,
                             uint32_t* res /*4, LDS*/) {
    constexpr int BPT = kHistBins / THREADS;
    constexpr int NW = THREADS / 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t h[BPT];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        h[i] = hist[t * BPT + i];
        sum += h[i];
    }
    // inclusive suffix scan over threads (thread THREADS-1 owns the top bins):
    // shuffles inside a wave, then the totals of the higher waves through LDS
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_down(incl, off, 64);
        if (lane + off < 64) incl += v;
    }
    if (lane == 0) s_part[w] = incl;
    __syncthreads();
    for (int ww = w + 1; ww < NW; ++ww) incl += s_part[ww];
    uint32_t above = incl - sum;  // count in bins owned by higher threads
    if (t == 0) {
        res[3] = incl;
        if (incl < k_rem) {  // fewer entries than requested: take them all
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

// ---- scan ------------------------------------------------------------------
struct ScanParams {
    const float* rows;
    uint32_t n, n_pad, dim;
    const float* q;
    float* scores;
    const uint32_t* keep;
    uint32_t mode;
    float thr;
    uint32_t nq;        // queries actually present (<= BQ: the pass may be padded; extra slots are never stored)
    uint32_t* work;     // work-queue head of this launch
    TaskTiers tiers;    // task t -> (first row, 64 / 32 / 16 rows)
    uint32_t n_tasks;   // tiers.total()
    float* gmax;        // [BQ][n_tasks] maximum valid score of each task's rows (-inf if none)
    uint64_t* gaux;     // nullable, [BQ][n_tasks]: (lane of that maximum << 32) | bits of the largest score of the task's OTHER rows
    unsigned long long* dbg;  // CQS_HIP_DEBUG_STAMPS: [16 + 2*wave] = start / end realtime of each wave
};

// NCH = ceil(dim / 256): 1-KiB chunks per row.  BQ queries, RI rows per batch (RI*BQ partial
// sums are reduced together).  FULL: dim == NCH*256.
// PIPE: 0 = batch by batch: loads -> math (many-query variants: the register file is full of query
//           fragments); 1 = two batches in flight while a third is reduced (tasks whose rows are partly
//           filtered out still go batch by batch, skipping the empty ones).
// OCC:  workgroups per CU the register allocation must leave room for.  The pipelined single-query
//       variant wants > 256 VGPRs (one wave per SIMD); corpora with fewer tasks than 2 waves per SIMD
//       use an OCC = 2 build instead so that every task is resident at once.
template <int NCH, int BQ, int RI, bool NT, bool FULL, int PIPE, int OCC>
__global__ __launch_bounds__(256, OCC) void scan_gemv_kernel(const ScanParams p) {
    constexpr int NV = RI * BQ;
    constexpr int LPV = 64 / NV;  // lanes per reduced value
    const int lane = threadIdx.x & 63;
    const uint32_t n = p.n, dim = p.dim;

    // per-lane column offsets: lane owns floats [c*256 + lane*4, +4) of every chunk c.
    // A partial last chunk is read from a clamped in-row address against a zero query
    // fragment, so the row loads stay unconditional (no branch, no early wait).
    uint32_t coff[NCH];
    f4 qv[BQ][NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t idx = (uint32_t)c * 256u + (uint32_t)lane * 4u;
        const bool in = FULL || idx < dim;
        coff[c] = in ? idx : dim - 4u;
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            const f4 v = *(const f4*)(p.q + (size_t)((uint32_t)b < p.nq ? b : 0) * dim + coff[c]);
            qv[b][c] = in ? v : (f4)(0.f);
        }
    }

    const uint32_t last = n - 1u;
    const uint32_t nwords = (n + 31u) / 32u;
    const uint32_t n_tasks = p.n_tasks;
    const uint32_t wpb = blockDim.x >> 6;  // waves per workgroup
    const uint32_t total_waves = gridDim.x * wpb;
    // (readfirstlane: tell the compiler the wave index - and every task index derived from it - is uniform)
    const uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
    // A wave's first task is its own index (no atomic).  A one-shot grid covers every task that way and
    // the hardware dispatcher does the scheduling; a persistent grid (huge corpora) continues from the
    // shared queue, whose tickets count from #waves.  (One queue word sustains only ~88 dequeues/us: a
    // start-up burst from every wave would cost 10-25 us.)
    const bool use_queue = n_tasks > total_waves;
    // A zero the compiler cannot see through.  With a provably uniform address hipcc rewrites the
    // dequeue into a wave-aggregated atomic followed at once by s_waitcnt vmcnt(0) + readfirstlane,
    // draining every row load in flight.  A "divergent" address keeps the plain returning atomic,
    // whose ticket is only waited for where it is used.
    uint32_t opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    if (p.dbg && lane == 0 && wave_id < kDbgWaves) p.dbg[16u + 2u * wave_id] = __builtin_amdgcn_s_memrealtime();

    // issue the RI*NCH row loads of batch j of the task at `base` back to back (all in flight together).
    // Address = uniform row pointer (SGPR pair, scalar ALU) + the lane's fixed 32-bit byte offset (+ the
    // chunk as an immediate): no per-row vector address registers.
    const char* const rows_b = (const char*)p.rows;
    const uint32_t row_bytes = dim * 4u;
    uint32_t lane_off[NCH];  // FULL: only [0] is used, chunks go into the immediate offset
#pragma unroll
    for (int c = 0; c < NCH; ++c) lane_off[c] = coff[c] * 4u;
    auto load_rows = [&](uint32_t base, int j, f4 (&x)[RI][NCH]) {
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            uint32_t row = base + (uint32_t)(RI * j + r);
            row = row > last ? last : row;
            const char* rp = rows_b + (uint64_t)row * row_bytes;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f4* src = FULL ? (const f4*)(rp + lane_off[0]) + c * 64 : (const f4*)(rp + lane_off[c]);
                if (NT) x[r][c] = __builtin_nontemporal_load(src);
                else x[r][c] = *src;
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the math that follows
    };
    // dot the batch with the queries; lane L = RI*j + r receives row r's score of query b in sc[b]
    auto reduce_rows = [&](int j, f4 (&x)[RI][NCH], float (&sc)[BQ]) {
        // packed f32 math (v_pk_fma_f32: two FMAs per instruction): every dot product keeps an (even, odd)
        // pair of partial sums over the lane's float pairs, folded once per batch
        f2 acc2[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc2[i] = (f2)(0.f);
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f2 xlo = __builtin_shufflevector(x[r][c], x[r][c], 0, 1);
                const f2 xhi = __builtin_shufflevector(x[r][c], x[r][c], 2, 3);
#pragma unroll
                for (int b = 0; b < BQ; ++b) {
                    f2 a = acc2[b * RI + r];
                    a = __builtin_elementwise_fma(xlo, __builtin_shufflevector(qv[b][c], qv[b][c], 0, 1), a);
                    a = __builtin_elementwise_fma(xhi, __builtin_shufflevector(qv[b][c], qv[b][c], 2, 3), a);
                    acc2[b * RI + r] = a;
                }
            }
        float acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = acc2[i].x + acc2[i].y;
        treduce<NV>(acc, lane);
        // value (b, r) now sits in lanes [(b*RI+r)*LPV, +LPV); lane L = RI*j + r wants it
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            const float t = __shfl(acc[0], (b * RI + (lane % RI)) * LPV, 64);
            if (lane / RI == j) sc[b] = t;
        }
    };

    // emit the scores of task `cur` (lane <-> row base+lane, lanes < trows): one coalesced store per
    // query and the task maximum for the select's pruning index
    auto epilogue = [&](uint32_t cur, uint32_t base, uint32_t trows, uint64_t mask, float (&sc)[BQ]) {
        const uint32_t row = base + (uint32_t)lane;
        const bool live = (uint32_t)lane < trows && ((mask >> lane) & 1ull);
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            float s = sc[b];
            // non-finite scores are never emitted (src/math.rs:23-27, src/cagra.rs:649-651)
            if (!live || !(__builtin_fabsf(s) <= 3.4028234664e38f)) s = -INFINITY;
            else if (p.mode == 1u) {
                // candidate.rs:550 clamp(0,1) (Rust clamp keeps -0.0), :513-519 `>= threshold`
                s = s < 0.f ? 0.f : (s > 1.f ? 1.f : s);
                if (!(s >= p.thr)) s = -INFINITY;
            }
            if ((uint32_t)b >= p.nq) continue;  // padding slot of a 5..7-query pass
            if ((uint32_t)lane < trows && row < p.n_pad) p.scores[(size_t)b * p.n_pad + row] = s;
            const float gm = wave_max64(s);
            if (lane == 0) p.gmax[(size_t)b * n_tasks + cur] = gm;
            if (p.gaux) {
                // Round 5: WHERE the maximum sits and the best score among the task's other rows.  A task whose runner-up
                // is below the select's threshold contributes exactly one candidate - (gm, base + arg) - without its score
                // row being read back (select_finish_kernel: 500 x 256 B of gather through ONE CU were 11 of its 28 us).
                // Rows tied at the maximum: only lane `arg` is masked, so the runner-up equals gm and the task is read.
                const uint32_t arg = (uint32_t)__builtin_ctzll(__ballot(s == gm));   // (no valid row: gm = -inf, every lane matches)
                const float sec = wave_max64(((uint32_t)lane == arg) ? -INFINITY : s);
                if (lane == 0) p.gaux[(size_t)b * n_tasks + cur] = ((uint64_t)arg << 32) | (uint64_t)__float_as_uint(sec);
            }
        }
    };
    // ticket -> task index
    auto claimed = [&](uint32_t ticket) -> uint32_t {
        return use_queue ? total_waves + (uint32_t)__builtin_amdgcn_readfirstlane(ticket) : n_tasks;
    };

    // wave-uniform mask of the rows of a task this wave must score: inside the corpus, kept by the filter
    auto task_mask = [&](uint32_t base, uint32_t trows) -> uint64_t {
        const uint64_t all = trows == 64u ? ~0ull : ((1ull << trows) - 1ull);
        uint64_t mask = all;
        if (base + trows > n) mask = (base >= n) ? 0ull : (all >> (trows - (n - base)));
        if (p.keep) {
            const uint32_t w = base / 32u;
            const uint32_t w0 = (w < nwords) ? p.keep[w] : 0u;
            const uint32_t w1 = (w + 1u < nwords) ? p.keep[w + 1u] : 0u;
            mask &= (((uint64_t)w1 << 32) | (uint64_t)w0) >> (base & 31u);
        }
        // uniform by construction; tell the compiler so the loop branches become scalar
        const uint32_t mlo = __builtin_amdgcn_readfirstlane((uint32_t)mask);
        const uint32_t mhi = __builtin_amdgcn_readfirstlane((uint32_t)(mask >> 32));
        return ((uint64_t)mhi << 32) | mlo;
    };
    f4 xa[RI][NCH], xb[RI][NCH];
    // batch by batch, skipping batches with no row to score
    auto sparse_task = [&](uint32_t t, uint32_t base, uint32_t trows, uint64_t mask) {
        float sc[BQ];
#pragma unroll
        for (int b = 0; b < BQ; ++b) sc[b] = -INFINITY;
        const int nb = (int)(trows / (uint32_t)RI);
        for (int j = 0; j < nb; ++j) {
            const uint32_t m = (uint32_t)(mask >> (RI * j)) & ((1u << RI) - 1u);
            if (m == 0u) continue;  // all RI rows filtered out / past the end: skip their HBM reads
            load_rows(base, j, xb);
            reduce_rows(j, xb, sc);
        }
        epilogue(t, base, trows, mask, sc);
    };

    uint32_t cur = wave_id;
    while (cur < n_tasks) {
        uint32_t trows;
        const uint32_t base = p.tiers.locate(cur, trows);
        const uint64_t mask = task_mask(base, trows);
        const uint64_t all = trows == 64u ? ~0ull : ((1ull << trows) - 1ull);
        uint32_t ticket = 0;  // lane 0: the dequeue drawn during this task
        if (PIPE == 1 && mask == all) {
            float sc[BQ];
#pragma unroll
            for (int b = 0; b < BQ; ++b) sc[b] = -INFINITY;
            const int steps = (int)(trows / (2u * (uint32_t)RI));
            load_rows(base, 0, xa);
            for (int s = 0; s < steps; ++s) {
                load_rows(base, 2 * s + 1, xb);
                // the dequeue goes out behind row loads already in flight: vmcnt retires in issue
                // order, so an atomic issued ahead of them would stall the first reduction
                if (s == 0 && use_queue && lane == 0) ticket = atomicAdd(p.work + opaque_zero, 1u);
                reduce_rows(2 * s, xa, sc);
                if (s + 1 < steps) load_rows(base, 2 * s + 2, xa);
                reduce_rows(2 * s + 1, xb, sc);
            }
            epilogue(cur, base, trows, mask, sc);
        } else {
            if (use_queue && lane == 0) ticket = atomicAdd(p.work + opaque_zero, 1u);
            sparse_task(cur, base, trows, mask);
        }
        cur = claimed(ticket);
    }
    if (p.dbg && lane == 0 && wave_id < kDbgWaves) p.dbg[17u + 2u * wave_id] = __builtin_amdgcn_s_memrealtime();
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

// ---- select: one workgroup per query -----------------------------------------
// Two-level exact top-k.  The scan left (a) every score and (b) the maximum of each
// 64-row group.  The workgroup histograms the group maxima (LDS), takes T = the highest
// bin with at least k maxima at or above it - then at least k scores have bin >= T, so
// every top-k entry has bin >= T and lives in a group whose maximum has bin >= T - reads
// back only those groups (about k of them), and sorts their entries with bin >= T
// (bitonic, packed keys).  Global loads are issued GB per thread at a time so the phases
// are bandwidth- not latency-paced.
constexpr uint32_t kGroupCap = 8192;
constexpr int kGB = 16;  // independent loads in flight per thread

__global__ __launch_bounds__(1024) void select_finish_kernel(const float* __restrict__ scores,
                                                             const float* __restrict__ gmax,
                                                             const uint64_t* __restrict__ gaux, uint32_t n_pad,
                                                             const TaskTiers tiers, uint32_t slot_log2, uint32_t k,
                                                             uint32_t row_base, uint32_t linear,
                                                             uint64_t* __restrict__ out_keys,
                                                             uint32_t* __restrict__ out_counts,
                                                             uint32_t* __restrict__ work,
                                                             unsigned long long* __restrict__ dbg) {
    __shared__ uint64_t s_keys[kCandCap];
    __shared__ uint32_t s_groups[kGroupCap];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[kHistBins];
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_res[4];
    __shared__ uint32_t s_cnt, s_ng, s_ng2;
    const uint32_t qi = blockIdx.x;
    const uint32_t n_tasks = tiers.total();
    const float* s = scores + (size_t)qi * n_pad;
    const float* gm = gmax + (size_t)qi * n_tasks;
    const int lane = threadIdx.x & 63;
    float r_lo = 0.f, r_scale = 0.f;                       // mode 2: the bins' range (see bin_of_range)

#define CQS_STAMP(i) do { if (dbg && threadIdx.x == 0 && blockIdx.x == 0) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    CQS_STAMP(0);
    for (uint32_t i = threadIdx.x; i < kHistBins; i += 1024u) s_hist[i] = 0u;
    if (threadIdx.x == 0) { s_cnt = 0; s_ng = 0; s_ng2 = 0; }
    __syncthreads();
    if (linear == 2u) {                                    // phase 0: smallest and largest finite group maximum
        float lo = INFINITY, hi = -INFINITY;
        for (uint32_t t0 = 0; t0 < n_tasks; t0 += 1024u * kGB) {   // kGB unconditional loads in flight (a conditional one per trip waited for each)
            float v[kGB];
#pragma unroll
            for (int u = 0; u < kGB; ++u) {
                const uint32_t t = t0 + (uint32_t)u * 1024u + threadIdx.x;
                v[u] = gm[t < n_tasks ? t : n_tasks - 1u];
            }
#pragma unroll
            for (int u = 0; u < kGB; ++u)
                if (v[u] != -INFINITY) { lo = fminf(lo, v[u]); hi = fmaxf(hi, v[u]); }   // (a clamped repeat of the last group changes neither)
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off, 64));
            hi = fmaxf(hi, __shfl_xor(hi, off, 64));
        }
        float* const f_part = reinterpret_cast<float*>(s_part);
        if (lane == 0) { f_part[threadIdx.x >> 6] = lo; f_part[16 + (threadIdx.x >> 6)] = hi; }
        __syncthreads();
        lo = f_part[0]; hi = f_part[16];
        for (int w = 1; w < 16; ++w) { lo = fminf(lo, f_part[w]); hi = fmaxf(hi, f_part[16 + w]); }
        __syncthreads();                                   // s_part is reused by block_decide
        r_lo = lo;
        r_scale = (hi > lo) ? 4096.0f / (hi - lo) : 0.f;   // (no finite maximum at all: nothing is histogrammed below)
        if (!(r_scale < INFINITY)) r_scale = 0.f;          // a range narrower than 4096 ulps of a subnormal: one bin, the exact path sorts it out
    }

    // phase 1: histogram of the group maxima
    float m[kGB];
    const bool one_pass = n_tasks <= 1024u * kGB;  // the maxima then stay in registers for phase 2
    for (uint32_t t0 = 0; t0 < n_tasks; t0 += 1024u * kGB) {
#pragma unroll
        for (int u = 0; u < kGB; ++u) {
            const uint32_t t = t0 + (uint32_t)u * 1024u + threadIdx.x;
            const float v = gm[t < n_tasks ? t : n_tasks - 1u];  // unconditional load, clamped
            m[u] = t < n_tasks ? v : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < kGB; ++u)
            if (m[u] != -INFINITY) atomicAdd(&s_hist[bin_any(m[u], linear, r_lo, r_scale)], 1u);
    }
    __syncthreads();
    CQS_STAMP(1);
    block_decide<1024>(s_hist, k, s_part, s_res);
    const uint32_t T = (s_res[3] < k) ? 0u : s_res[0];  // fewer groups than k: every valid score is a candidate
    __syncthreads();
    CQS_STAMP(2);

    // phase 2: groups whose maximum reaches the threshold bin
    for (uint32_t t0 = 0; t0 < n_tasks; t0 += 1024u * kGB) {
        if (!one_pass) {
#pragma unroll
            for (int u = 0; u < kGB; ++u) {
                const uint32_t t = t0 + (uint32_t)u * 1024u + threadIdx.x;
                const float v = gm[t < n_tasks ? t : n_tasks - 1u];
                m[u] = t < n_tasks ? v : -INFINITY;
            }
        }
        // list positions by ballot + mbcnt, one LDS atomic per wave (round 5; until then a 6-step shuffle scan over 16 flags)
        bool take[kGB];
        uint64_t mk[kGB];
        uint32_t pre[kGB], tot = 0;
#pragma unroll
        for (int u = 0; u < kGB; ++u) {
            take[u] = (m[u] != -INFINITY) && (bin_any(m[u], linear, r_lo, r_scale) >= T);
            mk[u] = __ballot(take[u]);
            pre[u] = tot;
            tot += (uint32_t)__popcll(mk[u]);
        }
        uint32_t base = 0;
        if (tot) {                                       // wave-uniform
            if (lane == 0) base = atomicAdd(&s_ng, tot);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        }
#pragma unroll
        for (int u = 0; u < kGB; ++u) {
            const uint32_t slot = base + pre[u] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[u], 0u));
            if (take[u] && slot < kGroupCap) s_groups[slot] = t0 + (uint32_t)u * 1024u + threadIdx.x;
        }
    }
    __syncthreads();
    CQS_STAMP(3);
    const uint32_t ng = s_ng;
    uint32_t count = kCandCap + 1u;
    if (ng <= kGroupCap) {
        // phase 3: their scores (L2 / Infinity Cache hits: the scan just wrote them).  Round 5: ONE WAVE PER GROUP -
        // lane <-> row of the group, so a group is one coalesced load at a wave-uniform base (no per-element
        // slot -> (group, offset) arithmetic), kGU groups in flight per wave, and the survivors are appended with
        // ballot + mbcnt ranks and ONE LDS atomic per kGU groups.  (Round 4 spread group x slot over all threads:
        // 2 rounds of 16 loads per thread with a 6-step shuffle scan each - 11 us of one CU's issue slots at k = 500.)
        // Round 5, producers that also left `gaux` (the gemv scan, the sparse index): a selected group whose runner-up
        // misses the threshold bin IS its maximum - the candidate (gm, base + arg) goes straight to the list and the
        // group's rows are never read; only groups with a second entry at or above the threshold (a few per cent at
        // k = 500, all of them under heavy ties) are gathered.
        const uint32_t* glist = s_groups;
        uint32_t n2 = ng;
        if (gaux) {
            uint32_t* const s_list2 = s_hist;              // the histogram is dead (T lives in a register)
            const uint64_t* const ga = gaux + (size_t)qi * n_tasks;
            for (uint32_t i0 = 0; i0 < ng; i0 += 1024u) {
                const uint32_t i = i0 + threadIdx.x;
                const bool valid = i < ng;
                const uint32_t t = s_groups[valid ? i : ng - 1u];
                const uint64_t ax = ga[t];
                const float mx = gm[t];
                const float sec = __uint_as_float((uint32_t)ax);
                const bool need = valid && (sec != -INFINITY) && (bin_any(sec, linear, r_lo, r_scale) >= T);
                const bool direct = valid && !need;
                const uint64_t mn = __ballot(need), md = __ballot(direct);
                uint32_t bn = 0, bd = 0;
                if (lane == 0) {
                    if (mn) bn = atomicAdd(&s_ng2, (uint32_t)__popcll(mn));
                    if (md) bd = atomicAdd(&s_cnt, (uint32_t)__popcll(md));
                }
                bn = (uint32_t)__builtin_amdgcn_readfirstlane((int)bn);
                bd = (uint32_t)__builtin_amdgcn_readfirstlane((int)bd);
                if (need) {
                    const uint32_t slot = bn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mn, 0u));
                    if (slot < kHistBins) s_list2[slot] = t;
                }
                if (direct) {
                    const uint32_t slot = bd + __builtin_amdgcn_mbcnt_hi((uint32_t)(md >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)md, 0u));
                    uint32_t grows;
                    const uint32_t gbase = tiers.locate(t, grows);
                    if (slot < kCandCap) s_keys[slot] = pack_key(okey(mx), row_base + gbase + (uint32_t)(ax >> 32));
                }
            }
            __syncthreads();
            if (s_ng2 <= kHistBins) { glist = s_list2; n2 = s_ng2; }
            else {                                         // more groups to read than the second list holds: read them all
                __syncthreads();
                if (threadIdx.x == 0) s_cnt = 0;
                __syncthreads();
            }
        }
        constexpr int kGU = 8;
        const uint32_t wv = threadIdx.x >> 6;
        for (uint32_t g0 = wv * kGU; g0 < n2; g0 += 16u * kGU) {
            float v[kGU];
            uint32_t idx[kGU];
            bool in[kGU];
#pragma unroll
            for (int u = 0; u < kGU; ++u) {
                const uint32_t g = g0 + (uint32_t)u;
                uint32_t grows;
                const uint32_t gbase = tiers.locate((uint32_t)__builtin_amdgcn_readfirstlane((int)glist[g < n2 ? g : n2 - 1u]), grows);
                in[u] = g < n2 && (uint32_t)lane < grows;
                idx[u] = gbase + (in[u] ? (uint32_t)lane : 0u);
            }
#pragma unroll
            for (int u = 0; u < kGU; ++u) v[u] = s[idx[u]];
            uint64_t mask[kGU];
            uint32_t pre[kGU], tot = 0;
            bool take[kGU];
#pragma unroll
            for (int u = 0; u < kGU; ++u) {
                take[u] = in[u] && (v[u] != -INFINITY) && (bin_any(v[u], linear, r_lo, r_scale) >= T);
                mask[u] = __ballot(take[u]);
                pre[u] = tot;
                tot += (uint32_t)__popcll(mask[u]);
            }
            uint32_t base = 0;
            if (tot) {                                   // wave-uniform
                if (lane == 0) base = atomicAdd(&s_cnt, tot);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            }
#pragma unroll
            for (int u = 0; u < kGU; ++u) {
                const uint32_t slot = base + pre[u] +
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask[u], 0u));
                if (take[u] && slot < kCandCap) s_keys[slot] = pack_key(okey(v[u]), row_base + idx[u]);
            }
        }
        __syncthreads();
        count = s_cnt;
    }
    if (count > kCandCap)  // heavy ties / crowded threshold bin: exact radix select over the whole row
        count = slow_select(s, n_pad, k, row_base, s_keys, s_hist, s_part, s_res, &s_cnt);

    CQS_STAMP(4);
    if (dbg && threadIdx.x == 0 && blockIdx.x == 0) { dbg[8] = ng; dbg[9] = count; }
    const uint64_t* sorted = s_keys;
    if (count <= 1024u) {
        // Rank sort: the keys are distinct, so rank(i) = #{j : key[j] > key[i]} is a permutation.
        // One key per thread, count broadcast LDS reads, no barriers inside the loop.
        // Round 5: the ranks are taken on the keys' TOP HALVES (the ordered score bits), four per 16-byte LDS read:
        // 1 read + 4 compares + 4 adds per four keys instead of 4 reads + 4 64-bit compares + 8 (7 -> ~3 us at 510 keys).
        // Two equal scores among the candidates get the same rank and leave a hole in the output - detected below, and
        // only then are the ranks retaken on the full keys (exact: scores equal in all 32 bits are duplicates or ties).
        uint64_t* s_sorted = reinterpret_cast<uint64_t*>(s_groups);  // group list is dead by now
        uint32_t* s_ok = s_hist;                                      // histogram is dead by now (slow_select included)
        __syncthreads();
        const uint32_t cpad = (count + 3u) & ~3u;
        if (threadIdx.x < cpad) s_ok[threadIdx.x] = threadIdx.x < count ? (uint32_t)(s_keys[threadIdx.x] >> 32) : 0u;
        if (threadIdx.x < count) s_sorted[threadIdx.x] = 0ull;       // (no valid key is 0)
        if (threadIdx.x == 0) s_res[0] = 0u;
        __syncthreads();
        if (threadIdx.x < count) {
            const uint64_t mine = s_keys[threadIdx.x];
            const uint32_t mh = (uint32_t)(mine >> 32);
            uint32_t rank = 0;
#pragma unroll 8
            for (uint32_t j = 0; j < cpad; j += 4u) {                 // (unrolled: eight 16-byte reads in flight, not one)
                const uint4 o = *reinterpret_cast<const uint4*>(&s_ok[j]);
                rank += (o.x > mh) ? 1u : 0u;
                rank += (o.y > mh) ? 1u : 0u;
                rank += (o.z > mh) ? 1u : 0u;
                rank += (o.w > mh) ? 1u : 0u;
            }
            s_sorted[rank] = mine;
        }
        __syncthreads();
        if (threadIdx.x < count && s_sorted[threadIdx.x] == 0ull) s_res[0] = 1u;   // a hole: two candidates share their score bits
        __syncthreads();
        if (s_res[0] != 0u) {
            if (threadIdx.x < count) {
                const uint64_t mine = s_keys[threadIdx.x];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < count; ++j) rank += (s_keys[j] > mine) ? 1u : 0u;
                s_sorted[rank] = mine;
            }
            __syncthreads();
        }
        sorted = s_sorted;
    } else {
        // Bitonic sort, descending (slow path sizes: up to kCandCap).
        uint32_t P = 2048;
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
    }
    CQS_STAMP(5);
    const uint32_t outc = count < k ? count : k;
    for (uint32_t i = threadIdx.x; i < k; i += 1024u) out_keys[(size_t)qi * k + i] = (i < outc) ? sorted[i] : 0ull;
    if (threadIdx.x == 0) out_counts[qi] = outc;
    // re-arm the scan work-queue heads for the next search (visible at the kernel boundary)
    if (qi == 0) for (uint32_t i = threadIdx.x; i < kWorkWords; i += 1024u) work[i] = 0u;
    CQS_STAMP(6);
#undef CQS_STAMP
}

// ---- launchers -------------------------------------------------------------
// dim <= 2048: up to 8 chunks per row, 1..8 queries per pass.  2052..4096 (the reference's presets reach 2560 and 4096,
// src/embedder/models.rs:515,572): 9..16 chunks per row - a row is 9..16 KiB, so ONE row per batch already keeps as many
// bytes in flight as eight 768-d rows; one query per pass (16 chunks x 4 registers of query + two row buffers = 192 VGPRs).
bool scan_dim_supported(uint32_t dim) { return dim >= 4 && dim % 4 == 0 && dim <= 4096; }

#ifndef CQS_SCAN_PIPE
#define CQS_SCAN_PIPE 1
#endif
#ifndef CQS_SCAN_RI1
#define CQS_SCAN_RI1 8   // rows per batch of the single-query scan
#endif
#ifndef CQS_SCAN_CAP_OCCUPANCY
#define CQS_SCAN_CAP_OCCUPANCY 1
#endif
#ifndef CQS_SCAN_ONE_SHOT
#define CQS_SCAN_ONE_SHOT 24u   // one task per wave up to this many tasks per SIMD; beyond: persistent grid + queue
#endif
#ifndef CQS_SCAN_BLOCK_WAVES
#define CQS_SCAN_BLOCK_WAVES 2u   // waves per workgroup of the one-shot launch (64-row tasks)
#endif
#ifndef CQS_SCAN_BLOCK_WAVES_SMALL
#define CQS_SCAN_BLOCK_WAVES_SMALL 1u   // ... of 16-row tasks
#endif
#ifndef CQS_SCAN_BLOCKS_PER_CU
#define CQS_SCAN_BLOCKS_PER_CU 2u   // persistent grid
#endif
#ifndef CQS_SCAN_TIER_B
#define CQS_SCAN_TIER_B 2u   // 32-row tasks at the end of the corpus, in units of (n_cu * 4 / 2)
#endif

// Task sizes over the padded corpus (measured at 768-d, 1 query; MI355X, 256 CUs):
//   < 8 64-row tasks per CU (<= 131k rows): 16-row tasks, so a 17.5k-row corpus still reaches every CU;
//   otherwise 64-row tasks (8 batches of 8 rows: the double buffer's fill/drain is amortised), with the
//   last two rounds' worth of 32-row tasks: the launch ends when its slowest wave does, and halving the
//   final tasks halves that tail (-5 us of 460 at 1M rows).  16-row tail tasks cost more than they gain.
TaskTiers plan_tiers(uint32_t n_pad, uint32_t n_cu, bool uniform64) {
    TaskTiers t{0u, 0u, 0u};
    const uint32_t n64 = n_pad / 64u, waves = n_cu * 4u;
    if (uniform64) { t.nA = n64; return t; }  // matrix-core kernel: 64-row groups only
    if (n64 < 8u * n_cu) { t.nC = n_pad / 16u; return t; }
    const uint32_t nb = CQS_SCAN_TIER_B * (waves / 2u) & ~1u;
    if (n64 >= 6u * waves && n64 > nb / 2u) { t.nA = n64 - nb / 2u; t.nB = nb; }
    else t.nA = n64;
    return t;
}
static uint32_t tier_slot_log2(const TaskTiers& t) { return t.nA ? 6u : (t.nB ? 5u : 4u); }

template <int NCH, int BQ, int RI>
static hipError_t launch_gemv(const ScanArgs& a, uint32_t q0, uint32_t nq, uint32_t work_slot, hipStream_t st) {
    constexpr int PIPE = (BQ <= 2) ? CQS_SCAN_PIPE : 0;
    ScanParams p;
    p.rows = a.rows; p.n = a.n; p.n_pad = a.n_pad; p.dim = a.dim;
    p.q = a.q + (size_t)q0 * a.dim;
    p.scores = a.scores + (size_t)q0 * a.n_pad;
    p.keep = a.keep; p.mode = a.mode; p.thr = a.threshold;
    p.nq = nq;
    p.work = a.work + work_slot;
    p.tiers = a.tiers;
    p.n_tasks = a.tiers.total();
    p.gmax = a.gmax + (size_t)q0 * p.n_tasks;
    p.gaux = a.gaux ? a.gaux + (size_t)q0 * p.n_tasks : nullptr;
    p.dbg = (unsigned long long*)a.dbg;
    // One-shot grid (one task per wave, the hardware dispatcher schedules: beats a persistent grid up to
    // ~1.5M rows) or, for huge corpora, a persistent grid that continues from the work queue (beats the
    // one-shot grid by 3 % at 10M rows).
    const bool small = a.tiers.nA == 0u && a.tiers.nB == 0u;  // 16-row tasks only
    const bool one_shot = p.n_tasks <= a.n_cu * 4u * CQS_SCAN_ONE_SHOT;
    uint32_t wpb = 4u, blocks = a.n_cu * CQS_SCAN_BLOCKS_PER_CU;
    if (one_shot) {
        wpb = small ? CQS_SCAN_BLOCK_WAVES_SMALL : CQS_SCAN_BLOCK_WAVES;
        blocks = (p.n_tasks + wpb - 1u) / wpb;
    }
    if (!one_shot && PIPE == 1 && CQS_SCAN_CAP_OCCUPANCY) blocks = a.n_cu;  // (one resident workgroup per CU, see below)
    const dim3 grid(blocks), block(64u * wpb);
    // Streaming a corpus far larger than the caches runs best with ONE wave per SIMD (each with two
    // 8-row batches in flight): a second wave per SIMD costs 2-3 % of the HBM rate (DRAM page
    // locality of the extra streams).  The pipelined kernel fits 2 waves per SIMD in registers, so
    // the launch asks for enough (unused) LDS to hold the CU at 4 waves.
    const bool huge = a.n_pad / 64u >= 6u * a.n_cu * 4u;  // >= ~393k rows (mid-size corpora prefer the extra waves)
    const size_t occ_lds = (PIPE == 1 && huge && CQS_SCAN_CAP_OCCUPANCY) ? (wpb == 4u ? 96u : 160u / (4u / wpb) - 16u) * 1024u : 0u;
    const bool full = (a.dim == (uint32_t)NCH * 256u);
#ifdef CQS_SCAN_FORCE_NT
    const bool nt = CQS_SCAN_FORCE_NT;
#else
    const bool nt = a.nontemporal;
#endif
#define CQS_LAUNCH(NTV, FULLV, OCCV) \
    do {                                                                                                     \
        auto kern = scan_gemv_kernel<NCH, BQ, RI, NTV, FULLV, PIPE, OCCV>;                                   \
        if (occ_lds > 64u * 1024u) {                                                                         \
            static DynLdsOnce once;   /* per instantiation: the attribute is set once per device, not per launch */ \
            hipError_t e = once.ensure((const void*)kern, occ_lds);                                          \
            if (e != hipSuccess) return e;                                                                   \
        }                                                                                                    \
        hipLaunchKernelGGL(kern, grid, block, occ_lds, st, p);                                               \
    } while (0)
    // small corpora never stream past the caches: their OCC = 2 variant is built without nt loads only
    if (small && PIPE == 1) { if (full) CQS_LAUNCH(false, true, 2); else CQS_LAUNCH(false, false, 2); }
    else if (nt) { if (full) CQS_LAUNCH(true, true, 1); else CQS_LAUNCH(true, false, 1); }
    else { if (full) CQS_LAUNCH(false, true, 1); else CQS_LAUNCH(false, false, 1); }
#undef CQS_LAUNCH
    return hipGetLastError();
}

template <int NCH>
static hipError_t launch_gemv_groups(const ScanArgs& a, hipStream_t st) {
    uint32_t done = 0, slot = 0;
    while (done < a.b) {
        const uint32_t left = a.b - done;
        hipError_t e;
        uint32_t g;
        // register budget ~ 4*NCH*(BQ + RI) + BQ*RI VGPRs: wide rows take fewer queries per pass
        if constexpr (NCH <= 4) {
            // 5..7 queries ride the 8-query pass (0.50 ms at 1M x 768; 4 + 1..3 would be two or three passes)
            if (left >= 5) { g = left < 8u ? left : 8u; e = launch_gemv<NCH, 8, 2>(a, done, g, slot, st); }
            else if (left >= 4) { g = 4; e = launch_gemv<NCH, 4, 4>(a, done, g, slot, st); }
            else if (left >= 2) { g = 2; e = launch_gemv<NCH, 2, 4>(a, done, g, slot, st); }
            else { g = 1; e = launch_gemv<NCH, 1, CQS_SCAN_RI1>(a, done, g, slot, st); }
        } else if constexpr (NCH <= 8) {
            if (left >= 2) { g = 2; e = launch_gemv<NCH, 2, 2>(a, done, g, slot, st); }
            else { g = 1; e = launch_gemv<NCH, 1, 2>(a, done, g, slot, st); }
        } else {
            g = 1; e = launch_gemv<NCH, 1, 1>(a, done, g, slot, st);
        }
        if (e != hipSuccess) return e;
        done += g;
        slot = (slot + 1u) % kWorkWords;
        if (slot == 0 && done < a.b) {  // queue heads exhausted: recycle them (rare: > 64 launches)
            e = hipMemsetAsync(a.work, 0, kWorkWords * sizeof(uint32_t), st);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

hipError_t launch_scan(const ScanArgs& a, hipStream_t st) {
    if (a.b == 0 || a.n == 0) return hipSuccess;
    if (!a.gemv_only && use_mfma(a.b, a.dim)) {
        uint32_t slot = 0;
        for (uint32_t q0 = 0; q0 < a.b; q0 += 256u) {
            const uint32_t nq = (a.b - q0) < 256u ? (a.b - q0) : 256u;
            hipError_t e = launch_scan_mfma(a, q0, nq, slot, st);
            if (e != hipSuccess) return e;
            slot = (slot + 1u) % kWorkWords;
            if (slot == 0 && q0 + 256u < a.b) {
                e = hipMemsetAsync(a.work, 0, kWorkWords * sizeof(uint32_t), st);
                if (e != hipSuccess) return e;
            }
        }
        return hipSuccess;
    }
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
        case 9: return launch_gemv_groups<9>(a, st);
        case 10: return launch_gemv_groups<10>(a, st);
        case 11: return launch_gemv_groups<11>(a, st);
        case 12: return launch_gemv_groups<12>(a, st);
        case 13: return launch_gemv_groups<13>(a, st);
        case 14: return launch_gemv_groups<14>(a, st);
        case 15: return launch_gemv_groups<15>(a, st);
        case 16: return launch_gemv_groups<16>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_select(const ScanArgs& a, uint32_t row_base, uint64_t* out_keys, uint32_t* out_counts,
                         hipStream_t st) {
    if (a.b == 0 || a.k == 0) return hipSuccess;
    hipLaunchKernelGGL(select_finish_kernel, dim3(a.b), dim3(1024), 0, st, a.scores, a.gmax,
                       (!a.gemv_only && use_mfma(a.b, a.dim)) ? nullptr : a.gaux, a.n_pad,
                       a.tiers, tier_slot_log2(a.tiers), a.k, row_base,
                       a.range_bins ? 2u : (a.linear_bins ? 1u : 0u), out_keys, out_counts, a.work,
                       (unsigned long long*)a.dbg);
    return hipGetLastError();
}

}  // namespace cqs
