// sharded.hip — ONE process, several GPUs: the row-sharded exact index behind the same C ABI handle.
//
// The reference is one process holding an `Arc<dyn VectorIndex>` (src/cli/batch/context.rs:157) whose GPU state
// sits behind one mutex (src/cagra.rs:263); `cqs_hip_index_create_sharded` gives that process the multi-GPU path
// of BASELINE.json's north_star without a second process: the corpus is cut row-wise (rowid order) into one
// shard per device, each shard is an ordinary single-device index (own stream, own scratch, `row_base` = its
// first global row), and one search is
//     host query block -> H2D to every device -> per-device scan + select (global row ids)
//     -> ONE RCCL all-gather over xGMI of the packed (score,row) keys (ncclGroupStart/End, one call per device)
//     -> D2H of the gathered lists from the first device -> host k-way merge (cqs_hip_merge_keys)
// with the comparator (score desc, row asc) being a total order on distinct keys, so the merged list is the
// single-device answer.  RCCL is bound at run time (dlopen: the library itself has no link-time
// dependency on it); a device list that names one device twice (the one-GPU test hook) cannot form an RCCL
// clique and gathers with device-to-device copies instead - same buffers, same merge.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>

#include "abi_guard.h"
#include "roctx.h"
#include "index_internal.h"

using namespace cqs_idx;

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
};

RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already in the process (PyTorch bundles one under the same soname) wins: one RCCL per process
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        api.lib = h;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(h, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
        api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    });
    return &api;
}

uint64_t popcount_bits(const uint32_t* words, uint64_t first_bit, uint64_t nbits) {   // first_bit % 32 == 0
    const uint32_t* w = words + first_bit / 32;
    uint64_t c = 0;
    for (uint64_t i = 0; i < nbits / 32; ++i) c += (uint64_t)__builtin_popcount(w[i]);
    if (nbits % 32) c += (uint64_t)__builtin_popcount(w[nbits / 32] & ((1u << (nbits % 32)) - 1u));
    return c;
}

}  // namespace

namespace cqs_sharded {

struct ShardSet {
    std::vector<cqs_hip_index*> shard;   // one single-device index per entry of the device list
    std::vector<uint64_t> lo;            // first row of each shard, relative to the parent's row_base
    bool use_rccl = false;               // distinct devices + RCCL loadable; else device-to-device copies
    std::vector<ncclComm_t> comm;
    std::vector<uint64_t*> d_gather;     // per shard, on its device: [n_shards][gather_cap] packed keys
    size_t gather_cap = 0;               // keys per shard slot (>= b * k of the current block)
    std::vector<hipEvent_t> ev;          // copy path: "shard s finished its select"
    uint64_t* h_gather = nullptr;        // pinned [n_shards][gather_cap]
    std::vector<uint64_t*> h_gather_dev; // copy path: h_gather's address as shard s's device sees it (null: not mappable)
    uint32_t* h_counts = nullptr;        // pinned [n_shards][count_cap]: the shards' per-query counts (copy path; unused by the merge)
    std::vector<uint32_t*> h_counts_dev;
    size_t count_cap = 0;
    float* h_q = nullptr;                // pinned query block
    size_t hq_cap = 0;                   // floats
    uint32_t searches = 0, gathers_rccl = 0;
};

namespace {

int32_t pfail(cqs_hip_index* p, int32_t code, const std::string& what) {
    p->last_error = what;
    if (code == CQS_HIP_ERR_DEVICE) p->poisoned.store(true, std::memory_order_release);
    return code;
}

// a child failed: surface its message on the parent
int32_t child_fail(cqs_hip_index* p, size_t s, int32_t rc) {
    cqs_hip_index* c = p->sh->shard[s];
    std::string msg;
    { std::lock_guard<std::mutex> g(c->mu); msg = c->last_error; }
    return pfail(p, rc, "shard " + std::to_string(s) + " (device " + std::to_string(c->device) + "): " + msg);
}

#define P_TRY(p, expr)                                                                                           \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess)                                                                                    \
            return pfail((p), _e == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE,                 \
                         std::string(#expr) + ": " + hipGetErrorString(_e));                                     \
    } while (0)

void free_gather(ShardSet* ss) {
    for (size_t s = 0; s < ss->d_gather.size(); ++s) {
        if (!ss->d_gather[s]) continue;
        (void)hipSetDevice(ss->shard[s]->device);
        (void)hipFree(ss->d_gather[s]);
        ss->d_gather[s] = nullptr;
    }
    if (ss->h_gather) { (void)hipHostFree(ss->h_gather); ss->h_gather = nullptr; }
    ss->h_gather_dev.clear();
    ss->gather_cap = 0;
}

int32_t ensure_gather(cqs_hip_index* p, size_t keys) {
    ShardSet* ss = p->sh;
    if (keys <= ss->gather_cap) return CQS_HIP_OK;
    for (cqs_hip_index* c : ss->shard) { P_TRY(p, hipSetDevice(c->device)); P_TRY(p, quiesce(c)); }
    free_gather(ss);
    const size_t G = ss->shard.size();
    ss->d_gather.assign(G, nullptr);
    for (size_t s = 0; s < G; ++s) {
        P_TRY(p, hipSetDevice(ss->shard[s]->device));
        P_TRY(p, hipMalloc(&ss->d_gather[s], G * keys * sizeof(uint64_t)));
    }
    P_TRY(p, hipHostMalloc((void**)&ss->h_gather, G * keys * sizeof(uint64_t), hipHostMallocPortable));
    // copy path (a device named twice): every shard's select kernel writes its keys straight into its slot of the
    // pinned gather buffer - no device-side gather buffer, no copy calls; any device that cannot map it keeps the copies
    ss->h_gather_dev.assign(G, nullptr);
    if (!ss->use_rccl) {
        for (size_t s = 0; s < G; ++s) {
            uint64_t* dp = nullptr;
            if (hipSetDevice(ss->shard[s]->device) != hipSuccess || hipHostGetDevicePointer((void**)&dp, ss->h_gather, 0) != hipSuccess) {
                (void)hipGetLastError();
                ss->h_gather_dev.assign(G, nullptr);
                break;
            }
            ss->h_gather_dev[s] = dp;
        }
    }
    ss->gather_cap = keys;
    return CQS_HIP_OK;
}

// pinned per-shard count rows for the direct-output copy path (the merge counts non-zero keys itself)
int32_t ensure_counts(cqs_hip_index* p, size_t nb) {
    ShardSet* ss = p->sh;
    const size_t G = ss->shard.size();
    if (nb <= ss->count_cap) return CQS_HIP_OK;
    for (cqs_hip_index* c : ss->shard) { P_TRY(p, hipSetDevice(c->device)); P_TRY(p, quiesce(c)); }
    if (ss->h_counts) (void)hipHostFree(ss->h_counts);
    ss->h_counts = nullptr; ss->count_cap = 0;
    ss->h_counts_dev.assign(G, nullptr);
    P_TRY(p, hipHostMalloc((void**)&ss->h_counts, G * nb * sizeof(uint32_t), hipHostMallocPortable));
    for (size_t s = 0; s < G; ++s) {
        uint32_t* dp = nullptr;
        if (hipSetDevice(ss->shard[s]->device) != hipSuccess || hipHostGetDevicePointer((void**)&dp, ss->h_counts, 0) != hipSuccess) {
            (void)hipGetLastError();
            ss->h_counts_dev.assign(G, nullptr);
            break;
        }
        ss->h_counts_dev[s] = dp;
    }
    ss->count_cap = nb;
    return CQS_HIP_OK;
}

int32_t ensure_hq(cqs_hip_index* p, size_t floats) {
    ShardSet* ss = p->sh;
    if (floats <= ss->hq_cap) return CQS_HIP_OK;
    if (ss->h_q) (void)hipHostFree(ss->h_q);
    ss->h_q = nullptr; ss->hq_cap = 0;
    P_TRY(p, hipHostMalloc((void**)&ss->h_q, floats * sizeof(float), hipHostMallocPortable));
    ss->hq_cap = floats;
    return CQS_HIP_OK;
}

// One block of `nb` staged queries (p->sh->h_q) through every shard, gathered and merged on the host.
// keep_host: nullable GLOBAL bitset (bit i = parent row i); bad[q] != 0: query q had a non-finite component.
int32_t search_block(cqs_hip_index* p, uint32_t nb, uint32_t k_eff, const uint32_t* keep_host, uint32_t mode, float thr,
                     const uint8_t* bad, uint64_t* out_rows, float* out_scores, uint32_t* out_counts, uint32_t k_out,
                     bool gemv_only = false) {
    ShardSet* ss = p->sh;
    const size_t G = ss->shard.size();
    const size_t keys = (size_t)nb * k_eff;
    int32_t rc = ensure_gather(p, keys);
    if (rc != CQS_HIP_OK) return rc;
    if (!ss->use_rccl && (rc = ensure_counts(p, nb)) != CQS_HIP_OK) return rc;
    // copy path, small blocks: the select kernels write into the pinned gather buffer themselves (slot s = [s * keys, + keys))
    const bool direct = !ss->use_rccl && keys <= kDirectOutKeys && !ss->h_gather_dev.empty() && ss->h_gather_dev[0] &&
                        !ss->h_counts_dev.empty() && ss->h_counts_dev[0];
    const size_t qbytes = (size_t)nb * p->dim * sizeof(float);
    // 1. every device: query block H2D, scan + select into its own d_out_keys ([nb][k_eff], zero padded)
    for (size_t s = 0; s < G; ++s) {
        cqs_hip_index* c = ss->shard[s];
        std::lock_guard<std::mutex> g(c->mu);
        if (c->poisoned.load(std::memory_order_acquire)) return child_fail(p, s, CQS_HIP_ERR_POISONED);
        P_TRY(p, hipSetDevice(c->device));
        if ((rc = ensure_scratch(c, nb, k_eff)) != CQS_HIP_OK) return child_fail(p, s, rc);
        if (c->done_valid && c->done_stream != c->stream) P_TRY(p, hipStreamWaitEvent(c->stream, c->done, 0));
        const uint32_t* d_keep = nullptr;
        bool skip = c->n == 0;
        if (!skip && keep_host) {
            const uint64_t inc = popcount_bits(keep_host, ss->lo[s], c->n);
            if (inc == 0) skip = true;                                  // nothing kept on this shard
            else if (inc < c->n) {
                if ((rc = stage_keep(c, keep_host + ss->lo[s] / 32, (c->n + 31) / 32)) != CQS_HIP_OK) return child_fail(p, s, rc);
                d_keep = c->d_keep;
            }
        }
        uint64_t* const keys_out = direct ? ss->h_gather_dev[s] + s * keys : c->d_out_keys;
        uint32_t* const counts_out = direct ? ss->h_counts_dev[s] + s * nb : c->d_out_counts;
        if (skip) {
            P_TRY(p, hipMemsetAsync(keys_out, 0, keys * sizeof(uint64_t), c->stream));
        } else {
            P_TRY(p, hipMemcpyAsync(c->d_q, ss->h_q, qbytes, hipMemcpyHostToDevice, c->stream));   // the query broadcast
            if ((rc = enqueue_search(c, c->d_q, nb, k_eff, d_keep, mode, thr, keys_out, counts_out, c->stream, gemv_only)) != CQS_HIP_OK)
                return child_fail(p, s, rc);
        }
    }
    // 2. gather every shard's keys on every device (RCCL all-gather over xGMI), or by copies into device 0's buffer
    CQS_ROCTX_RANGE("cqs_hip.shard_gather");
    if (ss->use_rccl) {
        RcclApi* api = rccl_api();
        ncclResult_t nr = api->GroupStart();
        for (size_t s = 0; nr == ncclSuccess && s < G; ++s) {
            cqs_hip_index* c = ss->shard[s];
            nr = api->AllGather(c->d_out_keys, ss->d_gather[s], keys, ncclUint64, ss->comm[s], c->stream);
        }
        const ncclResult_t ne = api->GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess)
            return pfail(p, CQS_HIP_ERR_DEVICE, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(nr) : "error"));
        ss->gathers_rccl++;
    } else if (direct) {
        // every select kernel has written its list into the pinned gather buffer itself: nothing is left to order on
        // device 0's stream, so the host waits for each shard's stream in turn - the first wait is the scan, the others
        // return at once.  (Until round 4 device 0's stream waited on one event per shard and the host on that stream:
        // G - 1 cross-stream barrier packets in front of the wake-up, 64 us between a query's last kernel and the next
        // query's first one against 28 us on a single-device handle - rocprofv3 kernel trace, a round-4 one-off script, since deleted.)
        for (size_t s = 0; s < G; ++s) {
            cqs_hip_index* c = ss->shard[s];
            P_TRY(p, hipSetDevice(c->device));
            P_TRY(p, hipStreamSynchronize(c->stream));
        }
    } else {
        cqs_hip_index* c0 = ss->shard[0];
        for (size_t s = 0; s < G; ++s) {
            cqs_hip_index* c = ss->shard[s];
            if (s != 0) {
                P_TRY(p, hipSetDevice(c->device));
                P_TRY(p, hipEventRecord(ss->ev[s], c->stream));
                P_TRY(p, hipSetDevice(c0->device));
                P_TRY(p, hipStreamWaitEvent(c0->stream, ss->ev[s], 0));
            } else {
                P_TRY(p, hipSetDevice(c0->device));
            }
            P_TRY(p, hipMemcpyAsync(ss->d_gather[0] + s * keys, c->d_out_keys, keys * sizeof(uint64_t), hipMemcpyDefault, c0->stream));
        }
    }
    // 3. one D2H from the first device and ONE host wait, on that device's stream: its copy of the gathered lists is
    // complete only when every shard has produced and sent its part (RCCL: the all-gather's receive side; copy path: the
    // per-shard events the copies waited on), i.e. when every shard's H2D of the pinned query block and its scan are
    // done too.  What the other devices still run (their own, unused, receive side) is ordered on THEIR streams before
    // anything the next search enqueues there; extend / save / destroy quiesce every stream themselves.  (Round 2 also
    // synchronised the other G - 1 streams here: G - 1 host round trips per query for nothing.)
    if (!direct) {
        cqs_hip_index* c0 = ss->shard[0];
        P_TRY(p, hipSetDevice(c0->device));
        P_TRY(p, hipMemcpyAsync(ss->h_gather, ss->d_gather[0], G * keys * sizeof(uint64_t), hipMemcpyDeviceToHost, c0->stream));
        P_TRY(p, hipStreamSynchronize(c0->stream));
    }
    // 4. host k-way merge per query (lists of query q: h_gather + s * keys + q * k_eff, zero padded)
    std::vector<uint32_t> counts(G);
    std::vector<uint64_t> merged(k_eff);
    for (uint32_t q = 0; q < nb; ++q) {
        if (bad[q]) { out_counts[q] = 0; continue; }                    // src/cagra.rs:464-470
        for (size_t s = 0; s < G; ++s) {
            const uint64_t* l = ss->h_gather + s * keys + (size_t)q * k_eff;
            uint32_t c = 0;
            while (c < k_eff && l[c] != 0) ++c;
            counts[s] = c;
        }
        const size_t m = cqs_hip_merge_keys(ss->h_gather + (size_t)q * k_eff, counts.data(), G, keys, k_eff, merged.data());
        cqs_hip_unpack_keys(merged.data(), m, out_rows + (size_t)q * k_out, out_scores + (size_t)q * k_out);
        out_counts[q] = (uint32_t)m;
    }
    ss->searches++;
    return CQS_HIP_OK;
}

}  // namespace

void destroy(cqs_hip_index* p) {
    ShardSet* ss = p->sh;
    for (cqs_hip_index* c : ss->shard) { (void)hipSetDevice(c->device); (void)quiesce(c); }
    if (ss->use_rccl) {
        RcclApi* api = rccl_api();
        for (ncclComm_t c : ss->comm) if (c) (void)api->CommDestroy(c);
    }
    free_gather(ss);
    for (size_t s = 0; s < ss->ev.size(); ++s)
        if (ss->ev[s]) { (void)hipSetDevice(ss->shard[s]->device); (void)hipEventDestroy(ss->ev[s]); }
    if (ss->h_q) (void)hipHostFree(ss->h_q);
    if (ss->h_counts) (void)hipHostFree(ss->h_counts);
    for (cqs_hip_index* c : ss->shard) cqs_hip_index_destroy(c);
    delete ss;
    p->sh = nullptr;
    delete p;
}

uint64_t len(const cqs_hip_index* p) {
    uint64_t n = 0;
    for (const cqs_hip_index* c : p->sh->shard) n += c->n;
    return n;
}

int32_t poisoned(const cqs_hip_index* p) {
    if (p->poisoned.load(std::memory_order_acquire)) return 1;
    for (const cqs_hip_index* c : p->sh->shard)
        if (c->poisoned.load(std::memory_order_acquire)) return 1;
    return 0;
}

size_t last_error(const cqs_hip_index* p, char* buf, size_t cap) {
    std::lock_guard<std::mutex> g(p->mu);
    const size_t m = p->last_error.size() < cap - 1 ? p->last_error.size() : cap - 1;
    memcpy(buf, p->last_error.data(), m);
    buf[m] = 0;
    return m;
}

void set_timing(cqs_hip_index* p, int32_t enable) {
    for (cqs_hip_index* c : p->sh->shard) cqs_hip_index_set_timing(c, enable);
}

// Scan time of the slowest shard per search is what the caller waits for; shards run concurrently, so report
// the first shard's count and the MAX of the shards' summed scan times.
int32_t scan_time(cqs_hip_index* p, uint32_t* launches, double* total_ms) {
    *launches = 0; *total_ms = 0.0;
    for (cqs_hip_index* c : p->sh->shard) {
        uint32_t l = 0; double ms = 0.0;
        const int32_t rc = cqs_hip_index_scan_time(c, &l, &ms);
        if (rc != CQS_HIP_OK) return rc;
        if (l > *launches) *launches = l;
        if (ms > *total_ms) *total_ms = ms;
    }
    return CQS_HIP_OK;
}

int32_t search(cqs_hip_index* p, const float* queries, uint32_t b, uint32_t query_dim, uint32_t k,
               const uint32_t* keep_bitset, uint32_t mode, float threshold, uint64_t* out_rows, float* out_scores,
               uint32_t* out_counts) {
    std::lock_guard<std::mutex> g(p->mu);
    if (poisoned(p)) return CQS_HIP_ERR_POISONED;
    if (b == 0) return CQS_HIP_OK;
    if (!queries || !out_counts) return pfail(p, CQS_HIP_ERR_INVALID, "search: null buffer");
    for (uint32_t i = 0; i < b; ++i) out_counts[i] = 0;
    const uint64_t n = len(p);
    if (n == 0 || k == 0) return CQS_HIP_OK;                       // src/cagra.rs:445-447
    if (query_dim != p->dim) {                                      // src/cagra.rs:449-456
        p->last_error = "search: query dimension mismatch (empty result)";
        return CQS_HIP_OK;
    }
    if (k > cqs::kMaxK) return pfail(p, CQS_HIP_ERR_INVALID, "search: k > max_k");
    if (mode > CQS_HIP_MODE_PIPELINE) return pfail(p, CQS_HIP_ERR_INVALID, "search: bad mode");
    if (!out_rows || !out_scores) return pfail(p, CQS_HIP_ERR_INVALID, "search: null output buffer");
    uint32_t k_eff = k;
    const uint32_t* keep = nullptr;
    if (keep_bitset) {                                              // src/cagra.rs:747-775, on the GLOBAL bitset
        const uint64_t included = popcount_bits(keep_bitset, 0, n);
        if (included == 0) return CQS_HIP_OK;
        if (included < n) {
            if (included < k_eff) k_eff = (uint32_t)included;
            keep = keep_bitset;
        }
    }
    uint32_t blk = 1024;
    for (const cqs_hip_index* c : p->sh->shard) blk = std::min(blk, max_query_block(c));
    std::vector<uint8_t> bad(b, 0);
    for (uint32_t done = 0; done < b;) {
        const uint32_t nb = std::min(b - done, blk);
        int32_t rc = ensure_hq(p, (size_t)nb * p->dim);
        if (rc != CQS_HIP_OK) return rc;
        for (uint32_t i = 0; i < nb; ++i) {
            const float* src = queries + (size_t)(done + i) * p->dim;
            float* dst = p->sh->h_q + (size_t)i * p->dim;
            bool ok = true;
            for (uint32_t d = 0; d < p->dim; ++d) ok &= std::isfinite(src[d]);
            bad[done + i] = !ok;
            if (ok) memcpy(dst, src, (size_t)p->dim * sizeof(float));
            else memset(dst, 0, (size_t)p->dim * sizeof(float));
        }
        rc = search_block(p, nb, k_eff, keep, mode, threshold, bad.data() + done, out_rows + (size_t)done * k,
                          out_scores + (size_t)done * k, out_counts + done, k);
        if (rc != CQS_HIP_OK) return rc;
        done += nb;
    }
    return CQS_HIP_OK;
}

// The combining queue's block on a sharded parent (index.hip, combine_lead): the callers' queries become ONE block
// through every shard (gemv passes only: a shard's scores do not depend on how many queries share its pass), one
// gather, one host merge per query; answers go back to each caller's own buffers.
int32_t search_combined(cqs_hip_index* p, cqs_combine_req* const* batch, uint32_t nb) {
    std::lock_guard<std::mutex> g(p->mu);
    if (poisoned(p)) return CQS_HIP_ERR_POISONED;
    if (p->inject_fail.exchange(0, std::memory_order_acq_rel) != 0)
        return pfail(p, CQS_HIP_ERR_DEVICE, "search: injected device failure (test hook)");
    const uint32_t k = batch[0]->k;
    for (uint32_t i = 0; i < nb; ++i) *batch[i]->out_count = 0;
    if (len(p) == 0) return CQS_HIP_OK;                              // src/cagra.rs:445-447
    uint32_t blk = 1024;
    for (const cqs_hip_index* c : p->sh->shard) blk = std::min(blk, max_query_block(c));
    static thread_local std::vector<uint64_t> rows;
    static thread_local std::vector<float> scores;
    static thread_local std::vector<uint32_t> counts;
    static thread_local std::vector<uint8_t> bad;
    rows.resize((size_t)nb * k); scores.resize((size_t)nb * k); counts.assign(nb, 0u); bad.assign(nb, 0);
    for (uint32_t done = 0; done < nb;) {
        const uint32_t b = std::min(nb - done, blk);
        int32_t rc = ensure_hq(p, (size_t)b * p->dim);
        if (rc != CQS_HIP_OK) return rc;
        for (uint32_t i = 0; i < b; ++i) memcpy(p->sh->h_q + (size_t)i * p->dim, batch[done + i]->q, (size_t)p->dim * sizeof(float));
        rc = search_block(p, b, k, nullptr, batch[0]->mode, batch[0]->thr, bad.data() + done, rows.data() + (size_t)done * k,
                          scores.data() + (size_t)done * k, counts.data() + done, k,
                          /*gemv_only=*/!(p->combine_relaxed && b >= cqs::kMfmaMinQueries));
        if (rc != CQS_HIP_OK) return rc;
        done += b;
    }
    for (uint32_t i = 0; i < nb; ++i) {
        memcpy(batch[i]->out_rows, rows.data() + (size_t)i * k, (size_t)counts[i] * sizeof(uint64_t));
        memcpy(batch[i]->out_scores, scores.data() + (size_t)i * k, (size_t)counts[i] * sizeof(float));
        *batch[i]->out_count = counts[i];
    }
    return CQS_HIP_OK;
}

// src/cli/commands/search/neighbors.rs:86-132 across shards: the target row comes back to the host from the
// shard that holds it (3 KB) and is searched like any query with k = limit + 1; the target is then dropped.
int32_t neighbors(cqs_hip_index* p, uint64_t target_row, uint32_t limit, uint64_t* out_rows, float* out_scores,
                  uint32_t* out_count) {
    std::lock_guard<std::mutex> g(p->mu);
    *out_count = 0;
    if (poisoned(p)) return CQS_HIP_ERR_POISONED;
    if (!out_rows || !out_scores) return pfail(p, CQS_HIP_ERR_INVALID, "neighbors: null output buffer");
    const uint64_t n = len(p);
    if (target_row < p->row_base || target_row - p->row_base >= n)
        return pfail(p, CQS_HIP_ERR_INVALID, "neighbors: target row not in this index");
    if (limit < 1u) limit = 1u;
    if (limit > CQS_HIP_NEIGHBORS_MAX) limit = CQS_HIP_NEIGHBORS_MAX;
    if (n <= 1) return CQS_HIP_OK;
    const uint32_t k = (uint64_t)limit + 1u < n ? limit + 1u : (uint32_t)n;
    ShardSet* ss = p->sh;
    int32_t rc = ensure_hq(p, p->dim);
    if (rc != CQS_HIP_OK) return rc;
    const uint64_t local = target_row - p->row_base;
    size_t owner = 0;
    for (size_t s = 0; s < ss->shard.size(); ++s)
        if (ss->shard[s]->n && local >= ss->lo[s] && local < ss->lo[s] + ss->shard[s]->n) owner = s;
    cqs_hip_index* c = ss->shard[owner];
    P_TRY(p, hipSetDevice(c->device));
    P_TRY(p, hipMemcpyAsync(ss->h_q, c->d_rows + (size_t)(local - ss->lo[owner]) * p->dim, (size_t)p->dim * sizeof(float),
                            hipMemcpyDeviceToHost, c->stream));
    P_TRY(p, hipStreamSynchronize(c->stream));
    std::vector<uint64_t> rows(k);
    std::vector<float> scores(k);
    uint32_t cnt = 0;
    const uint8_t bad = 0;
    rc = search_block(p, 1, k, nullptr, CQS_HIP_MODE_RAW, 0.f, &bad, rows.data(), scores.data(), &cnt, k);
    if (rc != CQS_HIP_OK) return rc;
    uint32_t outc = 0;
    for (uint32_t i = 0; i < cnt && outc < limit; ++i) {
        if (rows[i] == target_row) continue;                       // neighbors.rs:116-118
        out_rows[outc] = rows[i];
        out_scores[outc] = scores[i];
        ++outc;
    }
    *out_count = outc;
    return CQS_HIP_OK;
}

// Appends to the last shard that holds rows (rows stay contiguous in rowid order; the shards after it are empty
// and slide).  Balance returns at the next rebuild, like the reference's tiered backend (src/tiered.rs:1-43).
int32_t extend(cqs_hip_index* p, const float* rows, uint64_t n_new) {
    std::lock_guard<std::mutex> g(p->mu);
    if (poisoned(p)) return CQS_HIP_ERR_POISONED;
    if (n_new == 0) return CQS_HIP_OK;
    if (!rows) return pfail(p, CQS_HIP_ERR_INVALID, "extend: null rows");
    ShardSet* ss = p->sh;
    size_t t = 0;
    for (size_t s = 0; s < ss->shard.size(); ++s) if (ss->shard[s]->n) t = s;
    const int32_t rc = cqs_hip_index_extend(ss->shard[t], rows, n_new);
    if (rc != CQS_HIP_OK) return child_fail(p, t, rc);
    for (size_t s = t + 1; s < ss->shard.size(); ++s) {
        ss->lo[s] = ss->lo[t] + ss->shard[t]->n;
        ss->shard[s]->row_base = p->row_base + ss->lo[s];
    }
    return CQS_HIP_OK;
}

int32_t save(cqs_hip_index* p, const char* path, uint64_t* out_checksum) {
    std::lock_guard<std::mutex> g(p->mu);
    if (poisoned(p)) return CQS_HIP_ERR_POISONED;
    std::vector<Segment> segs;
    for (cqs_hip_index* c : p->sh->shard) {
        P_TRY(p, hipSetDevice(c->device));
        P_TRY(p, quiesce(c));
        if (c->n) segs.push_back(Segment{c->device, c->d_rows, c->n, c->stream});
    }
    return save_segments(p, segs, p->dim, p->metric, path, out_checksum);
}

// Shard plan: G near-equal contiguous row ranges whose starts are multiples of 256 rows (the score-row granule,
// and a whole number of keep-bitset words).
static std::vector<uint64_t> plan(uint64_t n, size_t G) {
    uint64_t per = (n + G - 1) / G;
    per = (per + 255) / 256 * 256;
    std::vector<uint64_t> lo(G + 1);
    for (size_t s = 0; s <= G; ++s) lo[s] = std::min<uint64_t>(n, per * s);
    return lo;
}

// Common tail of create / load: children exist with their rows; set up gather state and the RCCL clique.
static int32_t finish(cqs_hip_index* p, const int32_t* devices) {
    ShardSet* ss = p->sh;
    const size_t G = ss->shard.size();
    std::set<int> distinct(devices, devices + G);
    ss->ev.assign(G, nullptr);
    for (size_t s = 0; s < G; ++s) {
        P_TRY(p, hipSetDevice(ss->shard[s]->device));
        P_TRY(p, hipEventCreateWithFlags(&ss->ev[s], hipEventDisableTiming));
    }
    const char* force = getenv("CQS_HIP_SHARDED_GATHER");          // "copy" forces the copy path (A/B, debugging)
    if (distinct.size() == G && !(force && force[0] == 'c')) {
        RcclApi* api = rccl_api();
        if (api->ok()) {
            ss->comm.assign(G, nullptr);
            std::vector<int> devs(devices, devices + G);
            const ncclResult_t nr = api->CommInitAll(ss->comm.data(), (int)G, devs.data());
            if (nr != ncclSuccess)
                return pfail(p, CQS_HIP_ERR_DEVICE, std::string("ncclCommInitAll: ") + (api->GetErrorString ? api->GetErrorString(nr) : "error"));
            ss->use_rccl = true;
        } else if (G > 1) {
            return pfail(p, CQS_HIP_ERR_NO_DEVICE, "sharded index: librccl.so.1 not loadable (needed for more than one device)");
        }
    }
    return CQS_HIP_OK;
}

}  // namespace cqs_sharded

extern "C" {

int32_t cqs_hip_index_create_sharded(const float* rows, uint64_t n, uint32_t dim, uint32_t metric, const int32_t* devices,
                                     uint32_t n_devices, uint64_t row_base, cqs_hip_index** out) CQS_ABI_TRY {
    if (!out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64 || (n > 0 && !rows)) return CQS_HIP_ERR_INVALID;
    if (!cqs::scan_dim_supported(dim) || metric > CQS_HIP_METRIC_DOT) return CQS_HIP_ERR_INVALID;
    if (n + row_base > 0xFFFFFFFEull) return CQS_HIP_ERR_INVALID;
    cqs_hip_index* p = new (std::nothrow) cqs_hip_index();
    if (!p) return CQS_HIP_ERR_NOMEM;
    p->dim = dim; p->metric = metric; p->row_base = row_base; p->device = devices[0];
    cqs_idx::read_combine_env(p);
    p->sh = new (std::nothrow) cqs_sharded::ShardSet();
    if (!p->sh) { delete p; return CQS_HIP_ERR_NOMEM; }
    const std::vector<uint64_t> lo = cqs_sharded::plan(n, n_devices);
    int32_t rc = CQS_HIP_OK;
    for (uint32_t s = 0; s < n_devices && rc == CQS_HIP_OK; ++s) {
        cqs_hip_index* c = nullptr;
        rc = cqs_hip_index_create(rows ? rows + (size_t)lo[s] * dim : nullptr, lo[s + 1] - lo[s], dim, metric, devices[s],
                                  row_base + lo[s], &c);
        if (rc == CQS_HIP_OK) { p->sh->shard.push_back(c); p->sh->lo.push_back(lo[s]); }
    }
    if (rc == CQS_HIP_OK) rc = cqs_sharded::finish(p, devices);
    if (rc != CQS_HIP_OK) {
        if (getenv("CQS_HIP_VERBOSE")) fprintf(stderr, "[cqs_hip] create_sharded failed: %s\n", p->last_error.c_str());
        cqs_sharded::destroy(p);
        return rc;
    }
    *out = p;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

int32_t cqs_hip_index_load_sharded(const char* path, uint32_t expected_dim, uint64_t expected_rows, const int32_t* devices,
                                   uint32_t n_devices, uint64_t row_base, cqs_hip_index** out) CQS_ABI_TRY {
    if (!path || !out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) return CQS_HIP_ERR_INVALID;
    int fd = -1;
    uint64_t rows = 0, checksum = 0;
    uint32_t metric = 0;
    int32_t rc = open_blob(path, expected_dim, expected_rows, &fd, &rows, &metric, &checksum);
    if (rc != CQS_HIP_OK) return rc;
    if (rows + row_base > 0xFFFFFFFEull) { close(fd); return CQS_HIP_ERR_INVALID; }
    cqs_hip_index* p = new (std::nothrow) cqs_hip_index();
    if (!p) { close(fd); return CQS_HIP_ERR_NOMEM; }
    p->dim = expected_dim; p->metric = metric; p->row_base = row_base; p->device = devices[0];
    cqs_idx::read_combine_env(p);
    p->sh = new (std::nothrow) cqs_sharded::ShardSet();
    if (!p->sh) { close(fd); delete p; return CQS_HIP_ERR_NOMEM; }
    const std::vector<uint64_t> lo = cqs_sharded::plan(rows, n_devices);
    std::vector<Segment> segs;
    for (uint32_t s = 0; s < n_devices && rc == CQS_HIP_OK; ++s) {
        cqs_hip_index* c = nullptr;
        const uint64_t ns = lo[s + 1] - lo[s];
        rc = create_common(ns, expected_dim, metric, devices[s], row_base + lo[s], &c, &c);
        if (rc != CQS_HIP_OK) break;
        p->sh->shard.push_back(c);
        p->sh->lo.push_back(lo[s]);
        c->cap_rows = ns ? ns : 1;
        if (hipMalloc(&c->d_rows, (size_t)c->cap_rows * expected_dim * sizeof(float)) != hipSuccess) { rc = CQS_HIP_ERR_NOMEM; break; }
        if (ns) segs.push_back(Segment{devices[s], c->d_rows, ns, c->stream});
    }
    if (rc == CQS_HIP_OK) { rc = read_blob_into(fd, checksum, expected_dim, segs); fd = -1; }
    if (fd >= 0) close(fd);
    if (rc == CQS_HIP_OK) rc = cqs_sharded::finish(p, devices);
    if (rc != CQS_HIP_OK) { cqs_sharded::destroy(p); return rc; }
    *out = p;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

uint32_t cqs_hip_index_shards(const cqs_hip_index* x) CQS_ABI_TRY { return x ? (x->sh ? (uint32_t)x->sh->shard.size() : 1u) : 0u; } CQS_ABI_CATCH_VAL(0)

// Rows and device of shard s (a single-device handle is its own shard 0).  Returns CQS_HIP_ERR_INVALID past the end.
int32_t cqs_hip_index_shard_info(const cqs_hip_index* x, uint32_t s, int32_t* device, uint64_t* first_row, uint64_t* rows,
                                 int32_t* gathers_with_rccl) CQS_ABI_TRY {
    if (!x) return CQS_HIP_ERR_INVALID;
    if (!x->sh) {
        if (s != 0) return CQS_HIP_ERR_INVALID;
        if (device) *device = x->device;
        if (first_row) *first_row = x->row_base;
        if (rows) *rows = x->n;
        if (gathers_with_rccl) *gathers_with_rccl = 0;
        return CQS_HIP_OK;
    }
    if (s >= x->sh->shard.size()) return CQS_HIP_ERR_INVALID;
    const cqs_hip_index* c = x->sh->shard[s];
    if (device) *device = c->device;
    if (first_row) *first_row = c->row_base;
    if (rows) *rows = c->n;
    if (gathers_with_rccl) *gathers_with_rccl = x->sh->use_rccl ? 1 : 0;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

}  // extern "C"
