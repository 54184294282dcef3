// index_internal.h — the index handle and the host helpers shared by index.hip and sharded.hip.
// Internal to libcqs_hip.so (the public boundary is include/cqs_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cqs_hip.h"
#include "scan_kernels.h"

namespace cqs_sharded { struct ShardSet; }

// One single-query host search waiting for a pass over the corpus (the combining queue of cqs_hip_index_search).
// Lives on its caller's stack; the pointers are the caller's own output buffers.
struct cqs_combine_req {
    const float* q;          // [dim] host, validated (finite)
    uint32_t k, mode;
    float thr;
    uint64_t* out_rows;
    float* out_scores;
    uint32_t* out_count;
    int32_t rc = 0;
    bool done = false;
};

struct cqs_hip_index {
    int device = 0;
    uint64_t n = 0;         // rows
    uint64_t cap_rows = 0;  // allocated rows (owning index)
    uint32_t dim = 0;
    uint32_t metric = 0;
    uint64_t row_base = 0;
    bool borrow = false;
    float* d_rows = nullptr;
    hipStream_t stream = nullptr;

    // scratch, grown on demand (never inside an enqueue-only path once warm)
    uint32_t q_cap = 0;        // queries the scratch can hold
    uint64_t scr_n_pad = 0;    // score-row stride the scratch was sized for
    uint32_t k_cap = 0;
    float* d_q = nullptr;
    float* d_scores = nullptr;
    uint32_t* d_work = nullptr;   // scan work-queue heads
    unsigned long long* d_dbg = nullptr;  // CQS_HIP_DEBUG_STAMPS=1: select_finish phase stamps
    uint32_t n_cu = 256;
    float* d_gmax = nullptr;      // [q_cap, <= n_pad/16] per-task maxima (stride = tiers.total())
    uint64_t* d_gaux = nullptr;   // [min(q_cap, kGauxQueries), <= n_pad/16] (argmax lane, runner-up) of each task (gemv blocks)
    uint64_t* d_out_keys = nullptr;
    uint32_t* d_out_counts = nullptr;
    uint32_t* d_keep = nullptr;
    uint64_t keep_words_cap = 0;
    // pinned host staging
    float* h_q = nullptr;
    uint64_t* h_out_keys = nullptr;
    uint32_t* h_out_counts = nullptr;
    uint64_t* h_out_keys_dev = nullptr;     // device-visible addresses of the two buffers above (null: not mappable)
    uint32_t* h_out_counts_dev = nullptr;

    // Searches share one scratch (d_scores, d_gmax, d_work, d_q): the handle orders them across streams.
    // Every enqueue records `done` on its stream; an enqueue on a DIFFERENT stream first waits on it.
    hipEvent_t done = nullptr;
    hipStream_t done_stream = nullptr;
    bool done_valid = false;

    bool timing = false;
    std::vector<hipEvent_t> ev;  // pairs: [2i] before, [2i+1] after the scan launches
    size_t ev_used = 0;          // events recorded since the last read

    // Row-sharded parent (cqs_hip_index_create_sharded): the fields above are unused except dim / metric /
    // row_base; every entry point dispatches to sharded.hip.
    cqs_sharded::ShardSet* sh = nullptr;

    mutable std::mutex mu;
    std::atomic<bool> poisoned{false};
    std::string last_error;

    // Combining queue (index.hip, cqs_hip_index_search): concurrent single-query callers park here and ride ONE pass
    // over the corpus (up to kMaxGemvQ queries share the HBM stream in registers).  `cmu` orders the queue only; the
    // device work itself still runs under `mu`.  The reference serialises its callers behind Mutex<GpuState>
    // (src/cagra.rs:263) one search at a time; the daemon calls `search` from one thread per client
    // (src/cli/watch/daemon.rs:273).
    std::mutex cmu;
    std::condition_variable ccv;
    std::deque<cqs_combine_req*> pending;
    std::atomic<uint32_t> n_pending{0};   // = pending.size(), readable without cmu (the leader's short wait for stragglers)
    bool leader = false;                  // somebody is collecting / running a combined pass
    uint32_t expect = 1;                  // callers the next pass should expect (what recent passes saw); guarded by cmu
    bool combine = true;                  // CQS_HIP_COMBINE=0: every caller takes the serial path
    bool combine_relaxed = false;         // CQS_HIP_COMBINE_BITS=relaxed: blocks of >= 9 callers may run on the matrix cores (32 queries per
                                          // sweep instead of 8): answers within the parity tolerance of the lone call's, not its bits
    uint32_t combine_wait_us = 100;       // CQS_HIP_COMBINE_WAIT_US: how long after the END of a pass the next leader waits for the callers that pass carried
    std::chrono::steady_clock::time_point last_pass_end{};   // guarded by cmu (epoch until the first pass: nobody waits)
    std::atomic<uint64_t> stat_passes{0}, stat_queries{0};   // combined passes run / queries they carried
    std::atomic<int32_t> inject_fail{0};  // test hook (cqs_hip_debug_index_fail_next): the next host search fails as a device error
};

namespace cqs_idx {

constexpr size_t kMaxTimingEvents = 8192;
constexpr uint64_t kNtBytes = 200ull << 20;  // corpus larger than this streams past L2/MALL
constexpr uint32_t kGauxQueries = 32;        // query blocks up to this size (every gemv block the host paths form) carry the select's (argmax, runner-up) index
constexpr uint32_t kGauxMinK = 100;          // ... and only from this k on (below it the gather it replaces is a few groups)
constexpr size_t kDirectOutKeys = 8192;      // host searches of up to this many result keys have them written straight to pinned host memory

uint64_t pad_rows(uint64_t n);
int32_t fail(cqs_hip_index* idx, int32_t code, const char* what, hipError_t e = hipSuccess);
void free_scratch(cqs_hip_index* x);
int32_t ensure_scratch(cqs_hip_index* x, uint32_t b, uint32_t k);
uint32_t max_query_block(const cqs_hip_index* x);
// Enqueue scan + select for queries already on the device.  Caller holds mu.
int32_t enqueue_search(cqs_hip_index* x, const float* d_q, uint32_t b, uint32_t k, const uint32_t* d_keep,
                       uint32_t mode, float thr, uint64_t* d_out_keys, uint32_t* d_out_counts, hipStream_t st,
                       bool gemv_only = false);
hipError_t quiesce(cqs_hip_index* x);
int32_t stage_keep(cqs_hip_index* x, const uint32_t* host_words, uint64_t words);
int32_t create_common(uint64_t n, uint32_t dim, uint32_t metric, int32_t device, uint64_t row_base,
                      cqs_hip_index** out, cqs_hip_index** made);
void read_combine_env(cqs_hip_index* x);

// persistence over one or more device segments in row order (index.hip)
struct Segment { int device; float* d_rows; uint64_t rows; hipStream_t stream; };
int32_t save_segments(cqs_hip_index* err_owner, const std::vector<Segment>& segs, uint32_t dim, uint32_t metric,
                      const char* path, uint64_t* out_checksum);
int32_t open_blob(const char* path, uint32_t expected_dim, uint64_t expected_rows, int* fd_out, uint64_t* rows,
                  uint32_t* metric, uint64_t* checksum);
int32_t read_blob_into(int fd, uint64_t checksum, uint32_t dim, const std::vector<Segment>& segs);

}  // namespace cqs_idx

// Row-sharded parent handles (sharded.hip); each takes the parent handle and does its own locking.
namespace cqs_sharded {
void destroy(cqs_hip_index* parent);
int32_t search(cqs_hip_index* parent, const float* queries, uint32_t b, uint32_t query_dim, uint32_t k,
               const uint32_t* keep_bitset, uint32_t mode, float threshold, uint64_t* out_rows, float* out_scores,
               uint32_t* out_counts);
// One sealed block of the combining queue (index.hip): nb single-query callers with the same (k, mode, threshold), every
// query finite and of the parent's dimension; takes the parent mutex for the block.  gemv passes only, so each caller gets
// the bits its lone call gets.
int32_t search_combined(cqs_hip_index* parent, cqs_combine_req* const* batch, uint32_t nb);
int32_t neighbors(cqs_hip_index* parent, uint64_t target_row, uint32_t limit, uint64_t* out_rows, float* out_scores,
                  uint32_t* out_count);
int32_t extend(cqs_hip_index* parent, const float* rows, uint64_t n_new);
int32_t save(cqs_hip_index* parent, const char* path, uint64_t* out_checksum);
uint64_t len(const cqs_hip_index* parent);
int32_t poisoned(const cqs_hip_index* parent);
size_t last_error(const cqs_hip_index* parent, char* buf, size_t cap);
void set_timing(cqs_hip_index* parent, int32_t enable);
int32_t scan_time(cqs_hip_index* parent, uint32_t* launches, double* total_ms);
}  // namespace cqs_sharded

#define HIP_TRY(idx, expr)                                                        \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess)                                                     \
            return cqs_idx::fail((idx), _e == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE, #expr, _e); \
    } while (0)
