// scan_mfma.hip — batched scan: scores[q][row] for a block of 16..256 queries on the
// gfx950 matrix cores with exact-f32 MFMA (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit a k-ordered fmaf chain; 157 TF peak = the f32 vector rate).
//
// Reference counterpart: none in cqs itself (it scores one query per `VectorIndex::search`,
// src/index.rs:146); BASELINE.json configs[2] asks for "batch-256 query, LDS-tiled batched dot
// + per-query top-k".  Same outputs as scan_gemv_kernel: score rows (dropped = -inf) and one
// maximum per 64-row group; select_finish_kernel then takes the top k per query.
//
// Tiling (one workgroup = 4 waves = 1 per SIMD, 1 workgroup per CU, persistent + work queue):
//   workgroup tile  QT queries x RT corpus rows, K staged BK = 32 floats at a time
//   LDS             sQ[2][QT][36] + sR[2][RT][36] f32 (rows padded 32 -> 36 floats: the
//                   ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct
//                   4-bank slots -> conflict-free), double buffered, register staged
//                   (global_load_dwordx4 issued before the MFMA phase, ds_write_b128 after it)
//   wave tile       QW x RW MFMA tiles of 32x32 (acc = QW*RW*16 VGPRs)
//   MFMA operands   A = queries (i), B = corpus rows (j): lane l feeds A[i=l&31][k] and
//                   B[k][j=l&31] with k = 8*kg + 4*(l>>5) + c for the c-th of 4 MFMAs of a
//                   k-group - one ds_read_b128 per operand tile per 4 MFMAs.  Any k labelling
//                   is valid as long as A and B agree; this one makes both reads 16-B wide.
//   C layout        lane&31 = corpus row (so a score store is 128 B contiguous per register),
//                   register r <-> query (r&3) + 8*(r>>2) + 4*(lane>>5).
// Roofline: compute.  flops = 2*B*n*dim per batch (393 GFLOP at 256 x 1M x 768) against the
// 157.3 TF f32-MFMA peak; HBM traffic = corpus once + B*n*4 B of scores.
#include "scan_kernels.h"
#include "launch_util.h"

#include <cstdlib>
#include <type_traits>

namespace cqs {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;    // K floats per LDS stage
constexpr int kLDK = 36;   // padded LDS row stride in floats (144 B)

struct MfmaParams {
    const float* rows;
    uint32_t n, n_pad, dim;
    const float* q;      // [QT, dim] f32 (padded with zero rows up to QT)
    uint32_t b;          // live queries (<= QT)
    float* scores;       // [b, n_pad]
    float* gmax;         // [b, n_pad/64]
    const uint32_t* keep;
    uint32_t mode;
    float thr;
    uint32_t* work;
    uint32_t n_tasks;    // n_pad / RT  (scan_mfma16_kernel: x q_blocks)
    uint32_t q_blocks;   // scan_mfma16_kernel: query blocks of QT per row tile (task = row tile * q_blocks + query block)
};

// max over each 32-lane half of the wave, in every lane: four DPP steps inside the 16-lane rows (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror) + one v_permlane16_swap between the rows of a half - VALU only (as __shfl_xor these were
// 5 ds_bpermute round trips through the LDS queue per maximum, 160 per wave and tile)
template <int CTRL>
__device__ __forceinline__ float scan_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_wave_max(float m) {
    typedef unsigned pu2 __attribute__((ext_vector_type(2)));
    m = fmaxf(m, scan_dpp<0xB1>(m));
    m = fmaxf(m, scan_dpp<0x4E>(m));
    m = fmaxf(m, scan_dpp<0x141>(m));
    m = fmaxf(m, scan_dpp<0x140>(m));
    const pu2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}

template <int QW, int WQ, int RW, int WR, bool NT>
__global__ __launch_bounds__(64 * WQ * WR, WQ * WR / 4) void scan_mfma_kernel(const MfmaParams p) {
    constexpr int QT = 32 * QW * WQ, RT = 32 * RW * WR;
    constexpr int NTHR = 64 * WQ * WR;           // 4 waves (one per SIMD) or 8 (two per SIMD: while one wave waits on LDS or
                                                 // at the stage barrier its SIMD partner keeps the f32 matrix pipe fed)
    static_assert(WQ * WR == 4 || WQ * WR == 8, "4 or 8 waves per workgroup");
    static_assert(RW % 2 == 0, "a wave covers whole 64-row groups");
    constexpr int NF4 = (QT + RT) * (kBK / 4);   // float4 per stage
    constexpr int PER_T = NF4 / NTHR;            // float4 per thread per stage
    static_assert(NF4 % NTHR == 0, "stage divides over the workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sQ = smem;                            // [2][QT][kLDK]
    float* sR = smem + 2 * QT * kLDK;            // [2][RT][kLDK]
    __shared__ uint32_t s_task;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wq = wid / WR, wr = wid % WR;
    const int l31 = lane & 31, lh = lane >> 5;
    const uint32_t dim = p.dim, n = p.n;
    const uint32_t stages = dim / kBK;
    const uint32_t last_row = n - 1u;

    // staging map: float4 f of a stage -> (tile row, 16-B column); consecutive threads read
    // consecutive 16 B of one row: a full 128-B line per row per stage
    uint32_t st_row[PER_T], st_c4[PER_T];
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
        const uint32_t f = (uint32_t)u * (uint32_t)NTHR + (uint32_t)tid;
        st_row[u] = f >> 3;
        st_c4[u] = f & 7u;
    }

    // first tile = the workgroup's own index; later tiles from the shared queue (head counts from gridDim.x)
    uint32_t task = blockIdx.x;
    for (;; ) {
        if (task >= p.n_tasks) break;
        const uint32_t row0 = task * (uint32_t)RT;
        uint32_t next_task = 0;
        if (tid == 0) next_task = gridDim.x + atomicAdd(p.work, 1u);

        // global source pointers of this thread's staging slots (rows clamped inside the corpus)
        const float* src[PER_T];
#pragma unroll
        for (int u = 0; u < PER_T; ++u) {
            if (st_row[u] < (uint32_t)QT) {
                src[u] = p.q + (size_t)st_row[u] * dim + st_c4[u] * 4u;
            } else {
                uint32_t r = row0 + (st_row[u] - (uint32_t)QT);
                r = r > last_row ? last_row : r;
                src[u] = p.rows + (size_t)r * dim + st_c4[u] * 4u;
            }
        }
        auto stage_load = [&](uint32_t s, f4 (&reg)[PER_T]) {
#pragma unroll
            for (int u = 0; u < PER_T; ++u) {
                const float* a = src[u] + (size_t)s * kBK;
                if (NT && st_row[u] >= (uint32_t)QT) reg[u] = __builtin_nontemporal_load((const f4*)a);
                else reg[u] = *(const f4*)a;
            }
        };
        auto stage_write = [&](int buf, const f4 (&reg)[PER_T]) {
#pragma unroll
            for (int u = 0; u < PER_T; ++u) {
                float* dst = (st_row[u] < (uint32_t)QT)
                                 ? sQ + ((size_t)buf * QT + st_row[u]) * kLDK + st_c4[u] * 4u
                                 : sR + ((size_t)buf * RT + (st_row[u] - (uint32_t)QT)) * kLDK + st_c4[u] * 4u;
                *(f4*)dst = reg[u];
            }
        };

        f16v acc[QW][RW];
#pragma unroll
        for (int a = 0; a < QW; ++a)
#pragma unroll
            for (int b = 0; b < RW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        // One barrier per stage, placed between k-groups 2 and 3: the next stage's operands (global loads issued in
        // k-group 0, ds_writes in k-group 2, to the OTHER buffer) are visible after it, so k-group 3's MFMAs run while
        // the next stage's first fragments are read - the matrix pipe (one wave per SIMD: nobody else feeds it) no
        // longer drains at the stage boundary.  Within a k-group the loads / LDS operations are spread between the
        // MFMAs (sched_group_barrier) instead of in front of them.
        f4 reg[PER_T];
        stage_load(0, reg);
        stage_write(0, reg);
        __syncthreads();
        f4 af[2][QW], bf[2][RW];
        auto read_frags = [&](int buf, int kg, int set) {
            const float* aQ = sQ + ((size_t)buf * QT + (size_t)(wq * QW) * 32 + l31) * kLDK + 4 * lh;
            const float* aR = sR + ((size_t)buf * RT + (size_t)(wr * RW) * 32 + l31) * kLDK + 4 * lh;
#pragma unroll
            for (int a = 0; a < QW; ++a) af[set][a] = *(const f4*)(aQ + (size_t)a * 32 * kLDK + kg * 8);
#pragma unroll
            for (int b = 0; b < RW; ++b) bf[set][b] = *(const f4*)(aR + (size_t)b * 32 * kLDK + kg * 8);
        };
        read_frags(0, 0, 0);
        for (uint32_t s = 0; s < stages; ++s) {
            const int buf = (int)(s & 1u);
            const bool more = s + 1u < stages;
#pragma unroll
            for (int kg = 0; kg < kBK / 8; ++kg) {
                const int cur = kg & 1;
                if (kg == 0 && more) stage_load(s + 1u, reg);          // in flight under k-groups 0 and 1
                if (kg + 1 < kBK / 8) read_frags(buf, kg + 1, cur ^ 1);
                if (kg == kBK / 8 - 2 && more) stage_write(buf ^ 1, reg);
                if (kg == kBK / 8 - 1) {
                    __syncthreads();                                    // next stage's tile is in place
                    if (more) read_frags(buf ^ 1, 0, cur ^ 1);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int a = 0; a < QW; ++a)
#pragma unroll
                        for (int b = 0; b < RW; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a][c], bf[cur][b][c], acc[a][b], 0, 0, 0);
#ifndef CQS_MFMA_NO_SGB
                // one memory / LDS instruction after each MFMA while there are any (there are at most PER_T + QW + RW
                // + PER_T of them in a k-group and 4 QW RW MFMAs)
#pragma unroll
                for (int i = 0; i < 4 * QW * RW; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x0b0, 1, 0);   // VMEM read | DS (read or write)
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();   // (the last stage's barrier came before its final k-group: everyone is done reading LDS)

        // ---- epilogue: lane&31 <-> corpus row, register <-> query ----
        const uint32_t nwords = (n + 31u) / 32u;
        const uint32_t gpt = (uint32_t)RT / 64u;  // 64-row groups per tile
#pragma unroll
        for (int b = 0; b < RW; ++b) {
            const uint32_t row = row0 + (uint32_t)((wr * RW + b) * 32 + l31);
            bool live = row < n;
            if (p.keep && live) {
                const uint32_t w = row >> 5;
                live = w < nwords && ((p.keep[w] >> (row & 31u)) & 1u);
            }
#pragma unroll
            for (int a = 0; a < QW; ++a) {
                // 64-bit addressing (a 256-query block of a > 16.7M-row shard passes 2^32 score elements): the
                // per-lane part is one pointer, the per-register part (dq * n_pad) is wave-uniform
                const uint32_t q_lo = (uint32_t)((wq * QW + a) * 32 + 4 * lh);
                float* const sp = p.scores + (size_t)q_lo * p.n_pad + row;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t dq = (uint32_t)((r & 3) + 8 * (r >> 2));
                    float s = acc[a][b][r];
                    if (!live || !(__builtin_fabsf(s) <= 3.4028234664e38f)) s = -INFINITY;
                    else if (p.mode == 1u) {
                        s = s < 0.f ? 0.f : (s > 1.f ? 1.f : s);
                        if (!(s >= p.thr)) s = -INFINITY;
                    }
                    acc[a][b][r] = s;
                    if (q_lo + dq < p.b) sp[(size_t)dq * p.n_pad] = s;
                }
            }
        }
        // group maxima: a 64-row group = two adjacent 32-row MFMA tiles of this wave
#pragma unroll
        for (int g = 0; g < RW / 2; ++g)
#pragma unroll
            for (int a = 0; a < QW; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float m = fmaxf(acc[a][2 * g][r], acc[a][2 * g + 1][r]);
                    m = half_wave_max(m);
                    const uint32_t qi = (uint32_t)((wq * QW + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
                    if (l31 == 0 && qi < p.b)
                        p.gmax[(size_t)qi * (p.n_pad / 64u) + task * gpt + (uint32_t)(wr * (RW / 2) + g)] = m;
                }
        // the next tile was claimed at the top of this one (the atomic's round trip hides under the tile)
        if (tid == 0) s_task = next_task;
        __syncthreads();
        task = s_task;
        __syncthreads();   // s_task is free for the next hand-over
    }
}

// ---- the same tile with TWO workgroups per CU ---------------------------------------------------------------------------
// One workgroup per CU (above) leaves the matrix pipe idle whenever its four waves are all in the same non-MFMA phase:
// the tile epilogue (128 KB of score stores + 128 shuffle-maxima per lane: ~10 % of a 256 x 128 tile), the hand-over
// to the next tile, every stage barrier.  Two co-resident workgroups (one wave of each per SIMD) are in different phases
// most of the time.  For both to fit, K is staged 16 floats at a time (LDS 2 x 384 rows x 20 floats = 60 KB per
// workgroup; 20-float rows keep the 16 lanes of a ds_read_b128 cycle on 16 distinct 4-bank slots) and the registers
// stay under 256 (acc 128 + fragments 48 + two staging sets 48).  A stage is 2 k-groups of 8: the global loads of stage
// s + 2 are issued in k-group 0 of stage s into the register set that stage s - 1 has just written out; their data goes
// to LDS one stage later (a whole stage of MFMAs = 4 096 clocks covers the HBM latency).
constexpr int kBK16 = 16, kLDK16 = 20;

// OCC = workgroups per CU the register allocation leaves room for: 2 (the 128-query tiles: 64 q x 64 rows per wave), 3 for
// query blocks of <= 64 (<= 168 VGPRs; LDS 46-51 KB each): a 32 / 64-query block is HBM-bound or close to the ridge
// (2 B n dim flops over n dim 4 bytes = B / 2 flop per byte against ~23 at the ridge), and a third workgroup per CU keeps
// more of the corpus in flight under the two that multiply.
template <int QW, int WQ, int RW, int WR, bool NT, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void scan_mfma16_kernel(const MfmaParams p) {
    constexpr int QT = 32 * QW * WQ, RT = 32 * RW * WR;
    static_assert(WQ * WR == 4, "4 waves per workgroup");
    constexpr int NF4 = (QT + RT) * (kBK16 / 4);
    constexpr int PER_T = (NF4 + 255) / 256;       // (a stage that does not divide over the workgroup: the surplus slots re-copy the last float4)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sQ = smem;                            // [2][QT][kLDK16]
    float* sR = smem + 2 * QT * kLDK16;          // [2][RT][kLDK16]
    __shared__ uint32_t s_task;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wq = wid / WR, wr = wid % WR;
    const int l31 = lane & 31, lh = lane >> 5;
    const uint32_t dim = p.dim, n = p.n;
    const uint32_t stages = dim / kBK16;         // even (dim % 32 == 0)
    const uint32_t last_row = n - 1u;

    uint32_t st_row[PER_T], st_c4[PER_T];        // float4 f of a stage -> (tile row, 16-B column): 4 threads per row
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
        uint32_t f = (uint32_t)u * 256u + (uint32_t)tid;
        f = f < (uint32_t)NF4 ? f : (uint32_t)NF4 - 1u;
        st_row[u] = f >> 2;
        st_c4[u] = f & 3u;
    }

    uint32_t task = blockIdx.x;
    for (;; ) {
        if (task >= p.n_tasks) break;
        // consecutive tasks = the query blocks of ONE row tile: they run at about the same time on neighbouring workgroups,
        // so the tile's corpus rows come from HBM once and from L2 after that
        const uint32_t rtile = task / p.q_blocks, qblk = task % p.q_blocks;
        const uint32_t row0 = rtile * (uint32_t)RT;
        const float* const qbase = p.q + (size_t)qblk * QT * dim;
        const uint32_t b_live = p.b - qblk * (uint32_t)QT < (uint32_t)QT ? p.b - qblk * (uint32_t)QT : (uint32_t)QT;
        float* const scores = p.scores + (size_t)qblk * QT * p.n_pad;
        float* const gmax = p.gmax + (size_t)qblk * QT * (p.n_pad / 64u);
        uint32_t next_task = 0;
        if (tid == 0) next_task = gridDim.x + atomicAdd(p.work, 1u);
        const float* src[PER_T];
#pragma unroll
        for (int u = 0; u < PER_T; ++u) {
            if (st_row[u] < (uint32_t)QT) {
                src[u] = qbase + (size_t)st_row[u] * dim + st_c4[u] * 4u;
            } else {
                uint32_t r = row0 + (st_row[u] - (uint32_t)QT);
                r = r > last_row ? last_row : r;
                src[u] = p.rows + (size_t)r * dim + st_c4[u] * 4u;
            }
        }
        auto stage_load = [&](uint32_t s, f4 (&reg)[PER_T]) {
#pragma unroll
            for (int u = 0; u < PER_T; ++u) {
                const float* a = src[u] + (size_t)s * kBK16;
                if (NT && st_row[u] >= (uint32_t)QT) reg[u] = __builtin_nontemporal_load((const f4*)a);
                else reg[u] = *(const f4*)a;
            }
        };
        auto stage_write = [&](int buf, const f4 (&reg)[PER_T]) {
#pragma unroll
            for (int u = 0; u < PER_T; ++u) {
                float* dst = (st_row[u] < (uint32_t)QT)
                                 ? sQ + ((size_t)buf * QT + st_row[u]) * kLDK16 + st_c4[u] * 4u
                                 : sR + ((size_t)buf * RT + (st_row[u] - (uint32_t)QT)) * kLDK16 + st_c4[u] * 4u;
                *(f4*)dst = reg[u];
            }
        };
        f16v acc[QW][RW];
#pragma unroll
        for (int a = 0; a < QW; ++a)
#pragma unroll
            for (int b = 0; b < RW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        f4 regs[2][PER_T];
        stage_load(0, regs[0]);
        stage_load(1, regs[1]);
        stage_write(0, regs[0]);
        __syncthreads();
        f4 af[2][QW], bf[2][RW];
        auto read_frags = [&](int buf, int kg, int set) {
            const float* aQ = sQ + ((size_t)buf * QT + (size_t)(wq * QW) * 32 + l31) * kLDK16 + 4 * lh;
            const float* aR = sR + ((size_t)buf * RT + (size_t)(wr * RW) * 32 + l31) * kLDK16 + 4 * lh;
#pragma unroll
            for (int a = 0; a < QW; ++a) af[set][a] = *(const f4*)(aQ + (size_t)a * 32 * kLDK16 + kg * 8);
#pragma unroll
            for (int b = 0; b < RW; ++b) bf[set][b] = *(const f4*)(aR + (size_t)b * 32 * kLDK16 + kg * 8);
        };
        auto mma = [&](int cur) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int a = 0; a < QW; ++a)
#pragma unroll
                    for (int b = 0; b < RW; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a][c], bf[cur][b][c], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4 * QW * RW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x0b0, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        read_frags(0, 0, 0);
        // one stage: buffer `buf` multiplies; k-group 0 writes stage s + 1 (register set (s + 1) & 1, loaded a stage ago) into
        // the other buffer and requests stage s + 2 into the set stage s came from; k-group 1 crosses the barrier
        auto stage = [&](uint32_t s, auto par_c) {
            constexpr int par = decltype(par_c)::value;     // s & 1
            const bool more = s + 1u < stages;
            if (more) stage_write(par ^ 1, regs[par ^ 1]);
            if (s + 2u < stages) stage_load(s + 2u, regs[par]);
            read_frags(par, 1, 1);
            mma(0);
            __syncthreads();                                 // stage s + 1 is in place; everyone is done reading buffer par
            if (more) read_frags(par ^ 1, 0, 0);
            mma(1);
        };
        for (uint32_t s = 0; s < stages; s += 2u) {
            stage(s, std::integral_constant<int, 0>{});
            stage(s + 1u, std::integral_constant<int, 1>{});
        }
        __syncthreads();

        // ---- epilogue (as scan_mfma_kernel): lane&31 <-> corpus row, register <-> query ----
        const uint32_t nwords = (n + 31u) / 32u;
        const uint32_t gpt = (uint32_t)RT / 64u;
#pragma unroll
        for (int b = 0; b < RW; ++b) {
            const uint32_t row = row0 + (uint32_t)((wr * RW + b) * 32 + l31);
            bool live = row < n;
            if (p.keep && live) {
                const uint32_t w = row >> 5;
                live = w < nwords && ((p.keep[w] >> (row & 31u)) & 1u);
            }
#pragma unroll
            for (int a = 0; a < QW; ++a) {
                const uint32_t q_lo = (uint32_t)((wq * QW + a) * 32 + 4 * lh);
                float* const sp = scores + (size_t)q_lo * p.n_pad + row;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t dq = (uint32_t)((r & 3) + 8 * (r >> 2));
                    float sv = acc[a][b][r];
                    if (!live || !(__builtin_fabsf(sv) <= 3.4028234664e38f)) sv = -INFINITY;
                    else if (p.mode == 1u) {
                        sv = sv < 0.f ? 0.f : (sv > 1.f ? 1.f : sv);
                        if (!(sv >= p.thr)) sv = -INFINITY;
                    }
                    acc[a][b][r] = sv;
                    if (q_lo + dq < b_live) sp[(size_t)dq * p.n_pad] = sv;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < RW / 2; ++g)
#pragma unroll
            for (int a = 0; a < QW; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float m = fmaxf(acc[a][2 * g][r], acc[a][2 * g + 1][r]);
                    m = half_wave_max(m);
                    const uint32_t qi = (uint32_t)((wq * QW + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
                    if (l31 == 0 && qi < b_live)
                        gmax[(size_t)qi * (p.n_pad / 64u) + rtile * gpt + (uint32_t)(wr * (RW / 2) + g)] = m;
                }
        if (tid == 0) s_task = next_task;
        __syncthreads();
        task = s_task;
        __syncthreads();
    }
}

// ---- blocks of <= 64 queries: K split over the waves, the query tile resident in registers (round 5) --------------------
// What a small block wants is the one-query scan's access pattern (whole row pieces, contiguous, each fetched once) with the
// matrix cores' arithmetic.  The two kernels above stage EVERY operand through shared LDS tiles: a stage is 64 B (16 floats) of
// each of 256 rows plus the same 64 B of every query again, one workgroup barrier per stage, and HBM sees sector-sized pieces
// of 10^5 rows at once (4.5 TB/s at 32 queries against the 6.8 the one-query scan reads).  Here:
//   * a workgroup = 4 waves, 2 workgroups per CU; wave w owns the K quarter [w KQ, (w + 1) KQ) of EVERY row of the tile and of
//     every query: its 32 queries' fragments for that quarter live in registers for the whole launch (KG float4 = 96 VGPRs at
//     768 dimensions) - the query tile is read once per workgroup, not once per stage;
//   * the wave streams its quarter of the tile's rows in CHUNKS of 32 rows x 64 floats: per row 256 contiguous bytes, the three
//     chunks of a quarter 768 contiguous bytes fetched back to back; a chunk is loaded (8 x 16 B per lane, coalesced in 256-B
//     pieces) while the previous one multiplies, transposed through a wave-PRIVATE padded LDS image ([32][68] floats:
//     conflict-free ds_write_b128 / ds_read_b128) - no workgroup barrier anywhere in the K loop;
//   * per chunk: 8 fragment reads (one ds_read_b128 per 4 MFMAs; the A operand needs none) and 32 v_mfma_f32_32x32x2_f32;
//   * the four partial sums of a 32 q x 64 row tile meet once per tile through LDS (32 KB, over the chunk images), summed in
//     wave order; then the usual epilogue (lane & 31 <-> corpus row: 128-B score stores; one maximum per 64 rows).
// Blocks of 33-64 queries run as two query blocks of 32 on alternate workgroup octets (blocks b and b + 8 share an XCD's L2
// under round-robin placement, speed only) with a work queue each; the second reader of a row tile finds it in L2 / the
// Infinity Cache (default-policy loads).  dim % 256 == 0 and <= 1024 (the query registers); everything else keeps the
// kernels above.  Scores differ from theirs in summation order only (parity tolerance 1e-5, tests/test_scan_gpu.py).
constexpr int kKsRows = 64;        // corpus rows per tile (two MFMA row tiles)
constexpr int kKsLdr = 68;         // padded chunk row stride in floats (272 B: 16 rows -> 16 distinct 4-bank slots)

template <int KG, int WQ, bool NT>
__global__ __launch_bounds__(256 * WQ, 2) void scan_mfma_ks_kernel(const MfmaParams p) {
    constexpr int NCH = KG / 8;                  // 64-float chunks per row tile and wave
    static_assert(KG % 8 == 0, "a wave's K share is whole 64-float chunks");
    static_assert(WQ == 1 || WQ == 2, "one or two query tiles of 32");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 WQ][32][kKsLdr] chunk images; the combine reuses it as [WQ][4][32][64]
    __shared__ uint32_t s_task;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wq = wid >> 2, wk = wid & 3;       // query tile, K quarter of this wave
    const int l31 = lane & 31, lh = lane >> 5;
    const uint32_t dim = p.dim, n = p.n, last_row = n - 1u;
    float* const my = smem + (size_t)wid * 32 * kKsLdr;

    // this wave's query fragments: lane l feeds A[i = l & 31][k = wk KQ + 8 kg + 4 (l >> 5) + c] to the c-th MFMA of k-group kg
    f4 qa[KG];
    {
        // (K labelling: chunk ch of wave wk = floats [256 ch + 64 wk, + 64) - at every step the four waves' pieces of a row
        // are ADJACENT, 1 KB together, the one-query scan's request size; a contiguous quarter per wave read 256-B islands)
        const float* qp = p.q + ((size_t)wq * 32u + (uint32_t)l31) * dim + (uint32_t)wk * 64u + 4u * (uint32_t)lh;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) qa[kg] = *(const f4*)(qp + (kg / 8) * 256 + (kg % 8) * 8);
    }
    // chunk staging: float4 f = 64 u + lane of a chunk -> row f >> 4, 16-byte column f & 15 (a wave-load = 4 rows x 256 B)
    const uint32_t st_row0 = (uint32_t)lane >> 4, st_c4 = (uint32_t)lane & 15u;
    uint32_t task = blockIdx.x;                  // first tile = the workgroup's index, later ones from the queue
    const uint32_t n_tiles = p.n_tasks;

    auto chunk_load = [&](uint32_t row0, int st, f4 (&reg)[8]) {     // st = row tile (0, 1) * NCH + chunk
        const uint32_t rt = (uint32_t)st / NCH, ch = (uint32_t)st % NCH;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t r = row0 + rt * 32u + 4u * (uint32_t)u + st_row0;
            r = r > last_row ? last_row : r;
            const f4* a = (const f4*)(p.rows + (size_t)r * dim + ch * 256u + (uint32_t)wk * 64u + st_c4 * 4u);
            reg[u] = NT ? __builtin_nontemporal_load(a) : *a;
        }
    };
    // Two chunks in flight per wave (16 KB; 128 KB per CU at two workgroups): chunk st + 2 is requested as soon as chunk st has
    // moved from its registers to the LDS image.  The last two requests of a tile belong to the NEXT tile, so a workgroup
    // knows its next tile one tile ahead: the first two are static, the queue hands out the one after next.
    uint32_t task_next = task + gridDim.x;
#ifndef CQS_KS_DEPTH
#define CQS_KS_DEPTH 2
#endif
    constexpr int D = ((2 * NCH) % CQS_KS_DEPTH == 0 && 2 * NCH >= CQS_KS_DEPTH) ? CQS_KS_DEPTH : 2;   // chunks in flight per wave
    f4 reg[D][8];
    if (task < n_tiles) {
#pragma unroll
        for (int d = 0; d < D; ++d) chunk_load(task * (uint32_t)kKsRows, d, reg[d]);
    }
    while (task < n_tiles) {
        const uint32_t row0 = task * (uint32_t)kKsRows;
        // (past the end: request this tile's rows again - unconditional loads keep hipcc's wait counts those of the loop)
        const uint32_t row0_next = (task_next < n_tiles ? task_next : task) * (uint32_t)kKsRows;
        uint32_t claimed = 0;
        if (tid == 0) claimed = 2u * gridDim.x + atomicAdd(p.work, 1u);
        f16v acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int st = 0; st < 2 * NCH; ++st) {
            // the chunk requested two steps ago: registers -> this wave's LDS image (its reads of the previous chunk are
            // complete: their MFMAs have been issued), then the request for chunk st + 2 goes out under this one's MFMAs
#pragma unroll
            for (int u = 0; u < 8; ++u)
                *(f4*)(my + (4 * u + (int)st_row0) * kKsLdr + (int)st_c4 * 4) = reg[st % D][u];
            // (sched_barriers: left alone, hipcc sinks these loads to their first use two steps later and waits for them there
            // with vmcnt(0) - the whole HBM round trip exposed every other step; seen in the ISA, round 5)
            __builtin_amdgcn_sched_barrier(0);
            if (st + D < 2 * NCH) chunk_load(row0, st + D, reg[st % D]);
            else chunk_load(row0_next, st + D - 2 * NCH, reg[st % D]);
            __builtin_amdgcn_sched_barrier(0);
            const float* fr = my + l31 * kKsLdr + 4 * lh;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const f4 bf = *(const f4*)(fr + kg * 8);
                const f4 af = qa[(st % NCH) * 8 + kg];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[st / NCH] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c], bf[c], acc[st / NCH], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tid == 0) s_task = claimed;
        __syncthreads();                              // every wave is done with its chunk image; s_task is visible
        const uint32_t task_after = s_task;
        // ---- the four K quarters of a query tile meet: partial[wq][wk][t * 16 + r][lane], summed in wave order by the wave
        // that owns registers 4 wk .. 4 wk + 3 of both row tiles (it holds both halves of every 64-row group for the maximum) ----
        float* const part = smem + (size_t)wq * 4 * 32 * 64;      // 32 KB per query tile <= its four chunk images (34 816 B)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[((size_t)wk * 32 + t * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
        float sum[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * wk + j;
                float v = part[((size_t)0 * 32 + t * 16 + r) * 64 + lane];
                v += part[((size_t)1 * 32 + t * 16 + r) * 64 + lane];
                v += part[((size_t)2 * 32 + t * 16 + r) * 64 + lane];
                v += part[((size_t)3 * 32 + t * 16 + r) * 64 + lane];
                sum[t][j] = v;
            }
        __syncthreads();                              // the images are free for the next tile's chunks
        // ---- epilogue: lane & 31 <-> corpus row, register r <-> query (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
        const uint32_t nwords = (n + 31u) / 32u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t row = row0 + (uint32_t)(t * 32 + l31);
            bool live = row < n;
            if (p.keep && live) {
                const uint32_t w = row >> 5;
                live = w < nwords && ((p.keep[w] >> (row & 31u)) & 1u);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t qi = (uint32_t)(wq * 32 + j + 8 * wk + 4 * lh);       // r = 4 wk + j: (r & 3) = j, (r >> 2) = wk
                float sv = sum[t][j];
                if (!live || !(__builtin_fabsf(sv) <= 3.4028234664e38f)) sv = -INFINITY;
                else if (p.mode == 1u) {
                    sv = sv < 0.f ? 0.f : (sv > 1.f ? 1.f : sv);
                    if (!(sv >= p.thr)) sv = -INFINITY;
                }
                sum[t][j] = sv;
                if (qi < p.b) p.scores[(size_t)qi * p.n_pad + row] = sv;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m = half_wave_max(fmaxf(sum[0][j], sum[1][j]));
            const uint32_t qi = (uint32_t)(wq * 32 + j + 8 * wk + 4 * lh);
            if (l31 == 0 && qi < p.b) p.gmax[(size_t)qi * (p.n_pad / 64u) + task] = m;
        }
        task = task_next;
        task_next = task_after;
    }
}

template <int KG>
static hipError_t launch_mfma_ks(const ScanArgs& a, uint32_t q0, uint32_t nq, uint32_t slot, hipStream_t st) {
    MfmaParams p;
    p.rows = a.rows; p.n = a.n; p.n_pad = a.n_pad; p.dim = a.dim;
    p.q = a.q + (size_t)q0 * a.dim;
    p.b = nq;
    p.scores = a.scores + (size_t)q0 * a.n_pad;
    p.gmax = a.gmax + (size_t)q0 * (a.n_pad / kTaskRows);
    p.keep = a.keep; p.mode = a.mode; p.thr = a.threshold;
    p.work = a.work + slot;
    p.q_blocks = 1;
    p.n_tasks = a.n_pad / (uint32_t)kKsRows;
    // 33-64 queries: eight waves per workgroup, waves 4-7 hold the second query tile and stream the same row pieces as
    // waves 0-3 (the second fetch of a piece is an L1 / L2 hit: default-policy loads), one workgroup per CU; else two of four
    const bool two = nq > 32u;
    const size_t lds = (size_t)(two ? 8 : 4) * 32 * kKsLdr * sizeof(float);
    const uint32_t want = (two ? 1u : 2u) * a.n_cu;
    const uint32_t blocks = want < p.n_tasks ? want : p.n_tasks;
    const bool nt = a.nontemporal && !two;
    const void* kern;
    if (two) kern = (const void*)scan_mfma_ks_kernel<KG, 2, false>;
    else kern = nt ? (const void*)scan_mfma_ks_kernel<KG, 1, true> : (const void*)scan_mfma_ks_kernel<KG, 1, false>;
    static DynLdsOnce once[3];
    hipError_t e = once[two ? 2 : (nt ? 1 : 0)].ensure(kern, lds);
    if (e != hipSuccess) return e;
    void* args[] = {(void*)&p};
    return hipLaunchKernel(kern, dim3(blocks), dim3(two ? 512 : 256), args, lds, st);
}

// (Round 5 also built and removed a second form of the kernel above on v_mfma_f32_16x16x4_f32 tiles: 16 rows per step, so
// that a wave stages its WHOLE K quarter of them at once and a workgroup reads 48 KB of contiguous corpus per step (the
// one-query scan's stream) through XOR-swizzled unpadded images; one query tile of 16 for blocks of 9-16 (half the
// arithmetic, 164 VGPRs, three workgroups per CU).  Bit-correct, and exactly as fast: 0.5912 against 0.5889 ms at 32 queries,
// 0.5586 against 0.5598 at 16, 0.5502 against 0.5485 at 12 (same box, scan + select).  Neither the shape of the requests nor
// half the matrix work nor a third wave per SIMD moves this kernel: timing builds put it at 0.424 ms with no corpus loads
// after the first chunks, 0.545 with no score stores, 0.590 as built - profiles/r05_ksplit_ab.txt.  A third form staged the
// chunks by LDS-DMA (global_load_lds_dwordx4 from inline asm into two unpadded XOR-swizzled images per wave, counted
// vmcnt(8), the combine's partials in each wave's second image; 204 VGPRs): bit-correct, 0.5627 / 0.5598 against 0.5584 / 0.5661 ms
// at 32 queries, 0.5484 against 0.5262 at 16 - the register round trip is not what holds the stream back either; removed.)
// (Round 4 built and removed a STREAMING kernel for blocks of <= 32 queries - queries resident in LDS, every wave feeding
// v_mfma_f32_16x16x4_f32 with corpus rows loaded straight from global memory, 128 KB in flight per CU, no barrier in the
// loop: bit-correct, 0.70 ms at 32 queries x 1M rows with default-policy loads (0.80 streaming-policy; a 32x32x2 form with
// 32-byte pieces per row 1.13) against 0.68 for the LDS-tiled kernel below.  A lane-per-row operand layout reads 64 bytes
// of each of 64 rows per instruction: HBM sees ~10^5 concurrent row streams advancing one sector at a time and gives
// ~4.3 TB/s, where the one-query scan's 1-KiB-per-row loads get 6.8.  A block this small wants the gemv kernel's access
// pattern and the matrix cores' arithmetic; the transpose between the two is what the LDS stage is.)
template <int QW, int WQ, int RW, int WR, int OCC = 2>
static hipError_t launch_mfma16_cfg(const ScanArgs& a, uint32_t q0, uint32_t nq, uint32_t slot, hipStream_t st) {
    constexpr int QT = 32 * QW * WQ, RT = 32 * RW * WR;
    MfmaParams p;
    p.rows = a.rows; p.n = a.n; p.n_pad = a.n_pad; p.dim = a.dim;
    p.q = a.q + (size_t)q0 * a.dim;
    p.b = nq;
    p.scores = a.scores + (size_t)q0 * a.n_pad;
    p.gmax = a.gmax + (size_t)q0 * (a.n_pad / kTaskRows);
    p.keep = a.keep; p.mode = a.mode; p.thr = a.threshold;
    p.work = a.work + slot;
    p.q_blocks = (nq + (uint32_t)QT - 1u) / (uint32_t)QT;
    p.n_tasks = (a.n_pad / (uint32_t)RT) * p.q_blocks;
    const size_t lds = (size_t)2 * (QT + RT) * kLDK16 * sizeof(float);
    const uint32_t want = (uint32_t)OCC * a.n_cu;           // OCC workgroups per CU
    uint32_t blocks = want < p.n_tasks ? want : p.n_tasks;
    auto kern = a.nontemporal ? scan_mfma16_kernel<QW, WQ, RW, WR, true, OCC> : scan_mfma16_kernel<QW, WQ, RW, WR, false, OCC>;
    static DynLdsOnce once[2];
    hipError_t e = once[a.nontemporal ? 1 : 0].ensure((const void*)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, p);
    return hipGetLastError();
}

template <int QW, int WQ, int RW, int WR>
static hipError_t launch_mfma_cfg(const ScanArgs& a, uint32_t q0, uint32_t nq, uint32_t slot, hipStream_t st) {
    constexpr int QT = 32 * QW * WQ, RT = 32 * RW * WR;
    MfmaParams p;
    p.rows = a.rows; p.n = a.n; p.n_pad = a.n_pad; p.dim = a.dim;
    p.q = a.q + (size_t)q0 * a.dim;
    p.b = nq;
    p.scores = a.scores + (size_t)q0 * a.n_pad;
    p.gmax = a.gmax + (size_t)q0 * (a.n_pad / kTaskRows);   // the matrix-core path always uses 64-row groups
    p.keep = a.keep; p.mode = a.mode; p.thr = a.threshold;
    p.work = a.work + slot;
    p.n_tasks = a.n_pad / (uint32_t)RT;
    const size_t lds = (size_t)2 * (QT + RT) * kLDK * sizeof(float);
    uint32_t blocks = a.n_cu < p.n_tasks ? a.n_cu : p.n_tasks;
    auto kern = a.nontemporal ? scan_mfma_kernel<QW, WQ, RW, WR, true> : scan_mfma_kernel<QW, WQ, RW, WR, false>;
    static DynLdsOnce once[2];   // per instantiation x {nt, default}: set once per device, not per launch
    hipError_t e = once[a.nontemporal ? 1 : 0].ensure((const void*)kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WQ * WR), lds, st, p);
    return hipGetLastError();
}

// Query block [q0, q0+nq) (nq <= 256) through the MFMA kernel.  a.q must be readable (zero
// padded) up to the next multiple of the chosen query tile.
hipError_t launch_scan_mfma(const ScanArgs& a, uint32_t q0, uint32_t nq, uint32_t slot, hipStream_t st) {
    static const int waves = [] { const char* e = getenv("CQS_HIP_SCAN_MFMA_WAVES"); return e ? atoi(e) : 16; }();   // A/B hook: 4 = round-2 kernels, 8 = one 8-wave workgroup per CU
    static const int ksplit = [] { const char* e = getenv("CQS_HIP_SCAN_MFMA_KSPLIT"); return e ? atoi(e) : 1; }();  // A/B hook: 0 = the LDS-tiled kernels for every block size
    // blocks of <= 64 queries at 768 / 1024 dimensions: the K-split kernel (query tile in registers, 256-B row pieces)
    // (33-64 queries keep the LDS-tiled kernel: measured at 64 queries x 1M rows 1.02-1.05 ms on eight K-split waves per CU and
    // 1.13 ms with two query tiles per wave at one wave per SIMD, against 0.93 ms; CQS_HIP_SCAN_MFMA_KSPLIT=2 runs them anyway)
    if (ksplit && waves == 16 && nq <= (uint32_t)(ksplit == 2 ? 64 : 32)) {
        if (a.dim == 768u) return launch_mfma_ks<24>(a, q0, nq, slot, st);
        if (a.dim == 1024u) return launch_mfma_ks<32>(a, q0, nq, slot, st);
        if (a.dim == 512u) return launch_mfma_ks<16>(a, q0, nq, slot, st);
        if (a.dim == 256u) return launch_mfma_ks<8>(a, q0, nq, slot, st);
    }
    if (waves == 16) {                                       // 2 workgroups of 4 waves per CU, K staged 16 at a time
        if (nq > 64) return launch_mfma16_cfg<2, 2, 2, 2>(a, q0, nq, slot, st);   // 128 q x 128 rows (x 2 query blocks for 129-256)
        // Measured (1M x 768, scan + select, round 4): three workgroups per CU need <= 168 VGPRs - the 64-query tile then spills
        // 139 registers (1.05 ms against 0.94 at two per CU), the 32-query tile 20 (0.75 against 0.69); two per CU for both.
        // 9-32 queries used to take round 2's one-workgroup-per-CU kernel: 0.78 -> 0.69 ms at 32 queries, 0.77 -> 0.65 at 16.
#ifndef CQS_SCAN_MFMA_OCC64
#define CQS_SCAN_MFMA_OCC64 2
#endif
#ifndef CQS_SCAN_MFMA_OCC32
#define CQS_SCAN_MFMA_OCC32 2
#endif
        if (nq > 32) return launch_mfma16_cfg<2, 1, 2, 4, CQS_SCAN_MFMA_OCC64>(a, q0, nq, slot, st);   //  64 q x 256 rows
#if CQS_SCAN_MFMA_OCC32 > 0
        return launch_mfma16_cfg<1, 1, 2, 4, CQS_SCAN_MFMA_OCC32>(a, q0, nq, slot, st);                //  32 q x 256 rows
#endif
    }
    if (waves == 8) {
        if (nq > 128) return launch_mfma_cfg<2, 4, 2, 2>(a, q0, nq, slot, st);   // 256 q x 128 rows, 8 waves of 64 q x 64 rows
        if (nq > 64) return launch_mfma_cfg<2, 2, 2, 4>(a, q0, nq, slot, st);    // 128 q x 256 rows
        if (nq > 32) return launch_mfma_cfg<1, 2, 2, 4>(a, q0, nq, slot, st);    //  64 q x 256 rows (row tiles must divide n_pad: <= 256)
    }
    if (nq > 128) return launch_mfma_cfg<2, 4, 4, 1>(a, q0, nq, slot, st);   // 256 q x 128 rows
    if (nq > 64) return launch_mfma_cfg<2, 2, 2, 2>(a, q0, nq, slot, st);    // 128 q x 128 rows
    if (nq > 32) return launch_mfma_cfg<2, 1, 2, 4>(a, q0, nq, slot, st);    //  64 q x 256 rows
    return launch_mfma_cfg<1, 1, 2, 4>(a, q0, nq, slot, st);                 //  32 q x 256 rows
}

uint32_t mfma_query_tile(uint32_t nq) { return nq > 128 ? 256u : (nq > 64 ? 128u : (nq > 32 ? 64u : 32u)); }

}  // namespace cqs
