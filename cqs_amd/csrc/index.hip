// index.hip — host side of libcqs_hip.so: the exact GPU index behind the C ABI
// of include/cqs_hip.h.  Shape follows the reference's GPU backend exemplar
// `CagraIndex` (src/cagra.rs:255-277): a flat [n, dim] f32 dataset resident on
// the device, device work serialised behind one mutex (src/cagra.rs:263), a
// poisoned flag instead of panics (src/cagra.rs:472-489), stream sync before
// teardown (src/cagra.rs:289-302).  The id_map (row -> chunk id) stays with
// the caller (the Rust shim), as rows are addressed by integer here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <libgen.h>
#include <sys/stat.h>
#include <unistd.h>

#include "abi_guard.h"
#include "roctx.h"
#include "index_internal.h"
#include "persist_util.h"

using cqs::kMaxK;
using cqs::kRowsPerBlock;

namespace cqs_idx {

uint64_t pad_rows(uint64_t n) { return (n + kRowsPerBlock - 1) / kRowsPerBlock * kRowsPerBlock; }

int32_t fail(cqs_hip_index* idx, int32_t code, const char* what, hipError_t e) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    if (idx) {
        idx->last_error = buf;
        if (code == CQS_HIP_ERR_DEVICE) idx->poisoned.store(true, std::memory_order_release);
    }
    return code;
}

void free_scratch(cqs_hip_index* x) {
    hipFree(x->d_q); hipFree(x->d_scores); hipFree(x->d_gmax); hipFree(x->d_work); hipFree(x->d_gaux);
    x->d_work = nullptr; x->d_gaux = nullptr;
    hipFree(x->d_out_keys); hipFree(x->d_out_counts);
    hipHostFree(x->h_q); hipHostFree(x->h_out_keys); hipHostFree(x->h_out_counts);
    x->d_q = x->d_scores = nullptr; x->d_gmax = nullptr;
    x->d_out_keys = nullptr; x->d_out_counts = nullptr;
    x->h_q = nullptr; x->h_out_keys = nullptr; x->h_out_counts = nullptr;
    x->h_out_keys_dev = nullptr; x->h_out_counts_dev = nullptr;
    x->q_cap = 0; x->k_cap = 0; x->scr_n_pad = 0;
}

// Make the scratch hold `b` queries at top-`k` for the current n.
int32_t ensure_scratch(cqs_hip_index* x, uint32_t b, uint32_t k) {
    const uint64_t n_pad = pad_rows(x->n);
    if (b <= x->q_cap && k <= x->k_cap && n_pad == x->scr_n_pad) return CQS_HIP_OK;
    HIP_TRY(x, hipDeviceSynchronize());  // searches may be in flight on caller streams
    uint32_t qc = x->q_cap > b ? x->q_cap : b;
    uint32_t kc = x->k_cap > k ? x->k_cap : k;
    free_scratch(x);
    // query block, padded with zero rows to the MFMA query tile (<= 256 past the last chunk)
    HIP_TRY(x, hipMalloc(&x->d_q, ((size_t)qc + 256) * x->dim * sizeof(float)));
    HIP_TRY(x, hipMalloc(&x->d_scores, (size_t)qc * n_pad * sizeof(float)));
    HIP_TRY(x, hipMalloc(&x->d_work, cqs::kWorkWords * sizeof(uint32_t)));
    // work-queue heads must be zero on entry; every search re-zeroes them
    HIP_TRY(x, hipMemset(x->d_work, 0, cqs::kWorkWords * sizeof(uint32_t)));
    HIP_TRY(x, hipMalloc(&x->d_gmax, (size_t)qc * (n_pad / cqs::kTaskRowsSmall) * sizeof(float)));
    // (argmax, runner-up) per task: gemv blocks only (<= kGauxQueries queries; larger blocks run on the matrix cores or,
    // gemv_only, without it)
    HIP_TRY(x, hipMalloc(&x->d_gaux, (size_t)(qc < kGauxQueries ? qc : kGauxQueries) * (n_pad / cqs::kTaskRowsSmall) * sizeof(uint64_t)));
    HIP_TRY(x, hipMalloc(&x->d_out_keys, (size_t)qc * kc * sizeof(uint64_t)));
    HIP_TRY(x, hipMalloc(&x->d_out_counts, (size_t)qc * sizeof(uint32_t)));
    HIP_TRY(x, hipHostMalloc(&x->h_q, (size_t)qc * x->dim * sizeof(float), hipHostMallocDefault));
    HIP_TRY(x, hipHostMalloc(&x->h_out_keys, (size_t)qc * kc * sizeof(uint64_t), hipHostMallocDefault));
    HIP_TRY(x, hipHostMalloc(&x->h_out_counts, (size_t)qc * sizeof(uint32_t), hipHostMallocDefault));
    // device-visible addresses of the two result buffers (small blocks: the select kernel writes them directly);
    // a runtime that cannot map them leaves the pointers null and every block takes the copy path
    if (hipHostGetDevicePointer((void**)&x->h_out_keys_dev, x->h_out_keys, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&x->h_out_counts_dev, x->h_out_counts, 0) != hipSuccess) {
        (void)hipGetLastError();
        x->h_out_keys_dev = nullptr; x->h_out_counts_dev = nullptr;
    }
    x->q_cap = qc; x->k_cap = kc; x->scr_n_pad = n_pad;
    return CQS_HIP_OK;
}

// Largest query block whose score rows fit the scratch budget (score matrix
// is b * n_pad f32).  16 GiB default: 256 queries x 10M rows fit in one pass.
uint32_t max_query_block(const cqs_hip_index* x) {
    const uint64_t budget = 16ull << 30;
    const uint64_t per_q = pad_rows(x->n) * sizeof(float);
    uint64_t q = per_q ? budget / per_q : 1024;
    if (q < 1) q = 1;
    if (q > 1024) q = 1024;
    return (uint32_t)q;
}

// Enqueue scan + select for queries already on the device.  Caller holds mu.
int32_t enqueue_search(cqs_hip_index* x, const float* d_q, uint32_t b, uint32_t k, const uint32_t* d_keep,
                       uint32_t mode, float thr, uint64_t* d_out_keys, uint32_t* d_out_counts, hipStream_t st, bool gemv_only) {
    // the previous search may still be running on another stream and owns the same scratch
    if (x->done_valid && x->done_stream != st) HIP_TRY(x, hipStreamWaitEvent(st, x->done, 0));
    if (!gemv_only && cqs::use_mfma(b, x->dim)) {
        // the matrix-core path reads whole query tiles: stage the block in d_q with a zero tail
        const size_t qbytes = (size_t)b * x->dim * sizeof(float);
        if (d_q != x->d_q) HIP_TRY(x, hipMemcpyAsync(x->d_q, d_q, qbytes, hipMemcpyDeviceToDevice, st));
        HIP_TRY(x, hipMemsetAsync((char*)x->d_q + qbytes, 0, (size_t)256 * x->dim * sizeof(float), st));
        d_q = x->d_q;
    }
    cqs::ScanArgs a;
    a.rows = x->d_rows;
    a.n = (uint32_t)x->n;
    a.n_pad = (uint32_t)pad_rows(x->n);
    a.dim = x->dim;
    a.q = d_q;
    a.b = b;
    a.scores = x->d_scores;
    a.keep = d_keep;
    a.mode = mode;
    a.threshold = thr;
    a.nontemporal = x->n * x->dim * sizeof(float) > kNtBytes;
    a.linear_bins = (x->metric == CQS_HIP_METRIC_COSINE) || (mode == CQS_HIP_MODE_PIPELINE);
    a.k = k;
    a.gmax = x->d_gmax;
    static const bool use_gaux = [] { const char* e = getenv("CQS_HIP_SELECT_AUX"); return !(e && e[0] == '0'); }();   // A/B hook
    // Only where the gather it replaces is long: at k = 20 the index costs what it saves (same-box A/B, 1M x 768, scan + select per
    // step: k = 20 0.4698 with / 0.4667 ms without; k = 500 0.4783 / 0.4825 - tools/ab_select_aux.sh), so small k keeps round 4's path.
    a.gaux = (use_gaux && b <= kGauxQueries && k >= kGauxMinK) ? x->d_gaux : nullptr;
    a.work = x->d_work;
    a.n_cu = x->n_cu;
    a.dbg = x->d_dbg;
    a.gemv_only = gemv_only;
    a.tiers = cqs::plan_tiers(a.n_pad, x->n_cu, !gemv_only && cqs::uniform_groups(b, x->dim));
    const bool timed = x->timing && x->ev_used + 2 <= kMaxTimingEvents;
    if (timed) {
        while (x->ev.size() < x->ev_used + 2) {
            hipEvent_t e = nullptr;
            HIP_TRY(x, hipEventCreate(&e));
            x->ev.push_back(e);
        }
        HIP_TRY(x, hipEventRecord(x->ev[x->ev_used], st));
    }
    HIP_TRY(x, cqs::launch_scan(a, st));
    if (timed) {
        HIP_TRY(x, hipEventRecord(x->ev[x->ev_used + 1], st));
        x->ev_used += 2;
    }
    HIP_TRY(x, cqs::launch_select(a, (uint32_t)x->row_base, d_out_keys, d_out_counts, st));
    HIP_TRY(x, hipEventRecord(x->done, st));
    x->done_stream = st;
    x->done_valid = true;
    return CQS_HIP_OK;
}

// Wait (host) for the last enqueued search, whatever stream it ran on.  Caller holds mu.
hipError_t quiesce(cqs_hip_index* x) {
    hipError_t e = x->stream ? hipStreamSynchronize(x->stream) : hipSuccess;
    if (e == hipSuccess && x->done_valid) e = hipEventSynchronize(x->done);
    return e;
}

// Copy a host keep-bitset (`words` u32) into the handle's device copy on its stream.  Caller holds mu.
int32_t stage_keep(cqs_hip_index* x, const uint32_t* host_words, uint64_t words) {
    if (words > x->keep_words_cap) {
        HIP_TRY(x, quiesce(x));
        hipFree(x->d_keep);
        x->d_keep = nullptr;
        x->keep_words_cap = 0;
        HIP_TRY(x, hipMalloc(&x->d_keep, words * sizeof(uint32_t)));
        x->keep_words_cap = words;
    }
    HIP_TRY(x, hipMemcpyAsync(x->d_keep, host_words, words * sizeof(uint32_t), hipMemcpyHostToDevice, x->stream));
    return CQS_HIP_OK;
}

void read_combine_env(cqs_hip_index* x) {                                              // read once per handle
    if (const char* ce = getenv("CQS_HIP_COMBINE")) x->combine = ce[0] != '0';
    if (const char* cw = getenv("CQS_HIP_COMBINE_WAIT_US")) x->combine_wait_us = (uint32_t)atoi(cw);
    if (const char* cb = getenv("CQS_HIP_COMBINE_BITS")) x->combine_relaxed = cb[0] == 'r';
}

int32_t create_common(uint64_t n, uint32_t dim, uint32_t metric, int32_t device, uint64_t row_base,
                      cqs_hip_index** out, cqs_hip_index** made) {
    if (!out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    if (!cqs::scan_dim_supported(dim) || metric > CQS_HIP_METRIC_DOT) return CQS_HIP_ERR_INVALID;
    if (n + row_base > 0xFFFFFFFEull) return CQS_HIP_ERR_INVALID;  // row ids are packed in 32 bits
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return CQS_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= cnt) return CQS_HIP_ERR_INVALID;
    cqs_hip_index* x = new (std::nothrow) cqs_hip_index();
    if (!x) return CQS_HIP_ERR_NOMEM;
    x->device = device; x->n = n; x->dim = dim; x->metric = metric; x->row_base = row_base;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&x->done, hipEventDisableTiming) != hipSuccess) {
        if (x->stream) hipStreamDestroy(x->stream);
        delete x;
        return CQS_HIP_ERR_DEVICE;
    }
    read_combine_env(x);
    if (getenv("CQS_HIP_DEBUG_STAMPS")) {
        const size_t bytes = (16 + 2 * cqs::kDbgWaves) * sizeof(unsigned long long);
        if (hipMalloc(&x->d_dbg, bytes) == hipSuccess) (void)hipMemset(x->d_dbg, 0, bytes);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        x->n_cu = (uint32_t)prop.multiProcessorCount;
    *made = x;
    return CQS_HIP_OK;
}

}  // namespace cqs_idx

using namespace cqs_idx;

extern "C" {

const char* cqs_hip_version(void) CQS_ABI_TRY { return "cqs-hip 0.1.0 (gfx950)"; } CQS_ABI_CATCH_VAL("cqs-hip")

int32_t cqs_hip_device_count(void) CQS_ABI_TRY {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
} CQS_ABI_CATCH_NOHANDLE

int32_t cqs_hip_device_mem(int32_t device, uint64_t* free_bytes, uint64_t* total_bytes) CQS_ABI_TRY {
    if (hipSetDevice(device) != hipSuccess) return CQS_HIP_ERR_NO_DEVICE;
    size_t f = 0, t = 0;
    if (hipMemGetInfo(&f, &t) != hipSuccess) return CQS_HIP_ERR_DEVICE;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

int32_t cqs_hip_index_create(const float* rows, uint64_t n, uint32_t dim, uint32_t metric, int32_t device,
                             uint64_t row_base, cqs_hip_index** out) CQS_ABI_TRY {
    if (n > 0 && !rows) return CQS_HIP_ERR_INVALID;
    cqs_hip_index* x = nullptr;
    int32_t rc = create_common(n, dim, metric, device, row_base, out, &x);
    if (rc != CQS_HIP_OK) return rc;
    x->cap_rows = n ? n : 1;
    hipError_t e = hipMalloc(&x->d_rows, (size_t)x->cap_rows * dim * sizeof(float));
    if (e == hipSuccess && n)
        e = hipMemcpyAsync(x->d_rows, rows, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, x->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
    if (e != hipSuccess) {
        cqs_hip_index_destroy(x);
        return e == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE;
    }
    *out = x;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

int32_t cqs_hip_index_create_device(const void* d_rows, uint64_t n, uint32_t dim, uint32_t metric, int32_t device,
                                    uint64_t row_base, int32_t borrow, cqs_hip_index** out) CQS_ABI_TRY {
    if (n > 0 && !d_rows) return CQS_HIP_ERR_INVALID;
    if (((uintptr_t)d_rows & 15u) != 0) return CQS_HIP_ERR_INVALID;  // 16-B row loads
    cqs_hip_index* x = nullptr;
    int32_t rc = create_common(n, dim, metric, device, row_base, out, &x);
    if (rc != CQS_HIP_OK) return rc;
    if (borrow) {
        x->borrow = true;
        x->d_rows = (float*)d_rows;
        x->cap_rows = n;
    } else {
        x->cap_rows = n ? n : 1;
        hipError_t e = hipMalloc(&x->d_rows, (size_t)x->cap_rows * dim * sizeof(float));
        if (e == hipSuccess && n)
            e = hipMemcpyAsync(x->d_rows, d_rows, (size_t)n * dim * sizeof(float), hipMemcpyDeviceToDevice, x->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
        if (e != hipSuccess) {
            cqs_hip_index_destroy(x);
            return e == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE;
        }
    }
    *out = x;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

int32_t cqs_hip_index_extend(cqs_hip_index* x, const float* rows, uint64_t n_new) CQS_ABI_TRY {
    if (!x) return CQS_HIP_ERR_INVALID;
    if (x->sh) return cqs_sharded::extend(x, rows, n_new);
    std::lock_guard<std::mutex> g(x->mu);
    if (x->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    if (x->borrow) return fail(x, CQS_HIP_ERR_INVALID, "extend: index borrows its rows");
    if (n_new == 0) return CQS_HIP_OK;
    if (!rows) return fail(x, CQS_HIP_ERR_INVALID, "extend: null rows");
    if (x->n + n_new + x->row_base > 0xFFFFFFFEull) return fail(x, CQS_HIP_ERR_INVALID, "extend: row id overflow");
    HIP_TRY(x, hipSetDevice(x->device));
    HIP_TRY(x, quiesce(x));   // a search enqueued on a caller stream may still read d_rows
    const size_t row_bytes = (size_t)x->dim * sizeof(float);
    if (x->n + n_new > x->cap_rows) {  // grow geometrically, copy device-to-device
        uint64_t cap = x->cap_rows * 2;
        if (cap < x->n + n_new) cap = x->n + n_new;
        float* nd = nullptr;
        HIP_TRY(x, hipMalloc(&nd, cap * row_bytes));
        hipError_t e = hipMemcpyAsync(nd, x->d_rows, x->n * row_bytes, hipMemcpyDeviceToDevice, x->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
        if (e != hipSuccess) {
            hipFree(nd);
            return fail(x, CQS_HIP_ERR_DEVICE, "extend: copy", e);
        }
        hipFree(x->d_rows);
        x->d_rows = nd;
        x->cap_rows = cap;
    }
    HIP_TRY(x, hipMemcpyAsync(x->d_rows + x->n * x->dim, rows, n_new * row_bytes, hipMemcpyHostToDevice, x->stream));
    HIP_TRY(x, hipStreamSynchronize(x->stream));
    x->n += n_new;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(x)

}  // extern "C"

namespace {
using cqs_persist::Checksum;
using cqs_persist::exists;
using cqs_persist::fsync_parent;
using cqs_persist::read_all;
using cqs_persist::write_all;

struct FlatHeader {
    char magic[8];
    uint32_t version, dim, metric, pad;
    uint64_t rows, checksum;
    uint8_t reserved[24];
};
static_assert(sizeof(FlatHeader) == 64, "header is 64 bytes");
const char kFlatMagic[8] = {'C', 'Q', 'S', 'H', 'I', 'P', 'F', '1'};

constexpr size_t kIoPiece = 64ull << 20;   // pinned staging piece (x2: copy of piece i+1 overlaps file I/O of piece i)

// The rows of one or more device segments (row order) cut into pieces of <= 64 MiB that never straddle a segment.
struct Piece { const cqs_idx::Segment* seg; size_t off, len; };
std::vector<Piece> cut_pieces(const std::vector<cqs_idx::Segment>& segs, uint32_t dim) {
    std::vector<Piece> out;
    for (const cqs_idx::Segment& sg : segs) {
        const size_t bytes = (size_t)sg.rows * dim * sizeof(float);
        for (size_t off = 0; off < bytes; off += kIoPiece) out.push_back({&sg, off, bytes - off < kIoPiece ? bytes - off : kIoPiece});
    }
    return out;
}
struct PinPair {
    uint8_t* p[2] = {nullptr, nullptr};
    hipError_t alloc(size_t bytes) {
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc((void**)&p[i], bytes ? bytes : 8, hipHostMallocPortable);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    ~PinPair() { if (p[0]) hipHostFree(p[0]); if (p[1]) hipHostFree(p[1]); }
};
}  // namespace

namespace cqs_idx {

// Blob write = `save_blob_atomic_with_rollback` (src/cagra.rs:1468-1592): refuse on a stale `.bak`; stream the rows
// HBM -> pinned pieces -> `<path>.tmp` (checksummed on the way, fsync); move a live blob to `.bak`; rename tmp ->
// live; on failure restore `.bak`; on success drop it.  Host memory: two 64 MiB pinned pieces, whatever the corpus.
int32_t save_segments(cqs_hip_index* x, const std::vector<Segment>& segs, uint32_t dim, uint32_t metric, const char* path,
                      uint64_t* out_checksum) {
    const std::string live(path), bak = live + ".bak", tmp = live + ".tmp";
    if (exists(bak)) return fail(x, CQS_HIP_ERR_INVALID, "save: stale .bak from a prior failed save; manual recovery required");
    uint64_t rows = 0;
    for (const Segment& sg : segs) rows += sg.rows;
    const size_t bytes = (size_t)rows * dim * sizeof(float);
    const std::vector<Piece> pieces = cut_pieces(segs, dim);
    PinPair pin;
    hipError_t he = pin.alloc(bytes < kIoPiece ? bytes : kIoPiece);
    if (he != hipSuccess) return fail(x, CQS_HIP_ERR_NOMEM, "save: pinned staging", he);
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(x, CQS_HIP_ERR_INVALID, "save: cannot create temp file");
    FlatHeader h{};
    memcpy(h.magic, kFlatMagic, 8);
    h.version = 1; h.dim = dim; h.metric = metric; h.rows = rows;
    bool ok = write_all(fd, &h, sizeof h);   // checksum patched in below
    Checksum ck(bytes);
    auto issue = [&](size_t i) -> hipError_t {
        const Piece& pc = pieces[i];
        hipError_t e = hipSetDevice(pc.seg->device);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(pin.p[i & 1], (const uint8_t*)pc.seg->d_rows + pc.off, pc.len, hipMemcpyDeviceToHost, pc.seg->stream);
    };
    if (ok && !pieces.empty()) he = issue(0);
    for (size_t i = 0; ok && he == hipSuccess && i < pieces.size(); ++i) {
        he = hipStreamSynchronize(pieces[i].seg->stream);            // piece i is in pin[i & 1]
        if (he != hipSuccess) break;
        if (i + 1 < pieces.size() && (he = issue(i + 1)) != hipSuccess) break;
        ck.update(pin.p[i & 1], pieces[i].len, i + 1 == pieces.size());
        ok = write_all(fd, pin.p[i & 1], pieces[i].len);
    }
    if (pieces.empty()) ck.update(nullptr, 0, true);
    for (const Segment& sg : segs) (void)hipStreamSynchronize(sg.stream);
    h.checksum = ck.finish();
    ok = ok && he == hipSuccess && lseek(fd, 0, SEEK_SET) == 0 && write_all(fd, &h, sizeof h) && fsync(fd) == 0;
    close(fd);
    if (!ok) {
        unlink(tmp.c_str());
        return he != hipSuccess ? fail(x, CQS_HIP_ERR_DEVICE, "save: device copy", he) : fail(x, CQS_HIP_ERR_INVALID, "save: write failed");
    }
    const bool backed_up = exists(live);
    if (backed_up) {
        if (rename(live.c_str(), bak.c_str()) != 0) { unlink(tmp.c_str()); return fail(x, CQS_HIP_ERR_INVALID, "save: cannot back up the live blob"); }
        fsync_parent(live);
    }
    if (rename(tmp.c_str(), live.c_str()) != 0) {
        unlink(tmp.c_str());
        if (backed_up) {
            if (rename(bak.c_str(), live.c_str()) != 0) return fail(x, CQS_HIP_ERR_INVALID, "save failed and rollback failed: rename .bak back by hand");
            fsync_parent(live);
        }
        return fail(x, CQS_HIP_ERR_INVALID, "save: rename failed");
    }
    if (backed_up) unlink(bak.c_str());
    fsync_parent(live);
    if (out_checksum) *out_checksum = h.checksum;
    return CQS_HIP_OK;
}

// `CagraIndex::load` (src/cagra.rs:1174-1330), part 1: header / size checks.  Leaves the file open at the rows.
int32_t open_blob(const char* path, uint32_t expected_dim, uint64_t expected_rows, int* fd_out, uint64_t* rows,
                  uint32_t* metric, uint64_t* checksum) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return CQS_HIP_ERR_INVALID;
    FlatHeader h{};
    struct stat st;
    const bool ok = read_all(fd, &h, sizeof h) && memcmp(h.magic, kFlatMagic, 8) == 0 && h.version == 1 &&
                    h.dim == expected_dim && (expected_rows == 0 || h.rows == expected_rows) && h.metric <= CQS_HIP_METRIC_DOT &&
                    fstat(fd, &st) == 0 && h.dim != 0 && h.rows <= (UINT64_MAX - sizeof h) / ((uint64_t)h.dim * 4u) &&
                    (uint64_t)st.st_size == sizeof h + h.rows * h.dim * 4u;
    if (!ok) { close(fd); return CQS_HIP_ERR_INVALID; }
    *fd_out = fd; *rows = h.rows; *metric = h.metric; *checksum = h.checksum;
    return CQS_HIP_OK;
}

// part 2: the rows stream file -> pinned pieces -> HBM (the segments' buffers are allocated by the caller) while
// the checksum is recomputed.  Closes fd.  A mismatch returns CQS_HIP_ERR_INVALID and the caller discards the index.
int32_t read_blob_into(int fd, uint64_t checksum, uint32_t dim, const std::vector<Segment>& segs) {
    uint64_t rows = 0;
    for (const Segment& sg : segs) rows += sg.rows;
    const size_t bytes = (size_t)rows * dim * sizeof(float);
    const std::vector<Piece> pieces = cut_pieces(segs, dim);
    PinPair pin;
    hipError_t he = pin.alloc(bytes < kIoPiece ? bytes : kIoPiece);
    Checksum ck(bytes);
    bool ok = true;
    for (size_t i = 0; ok && he == hipSuccess && i < pieces.size(); ++i) {
        const Piece& pc = pieces[i];
        if (i >= 2) he = hipStreamSynchronize(pieces[i - 2].seg->stream);   // pin[i & 1] was the source of piece i-2's copy
        if (he != hipSuccess) break;
        ok = read_all(fd, pin.p[i & 1], pc.len);                            // overlaps the H2D copy of piece i-1
        if (!ok) break;
        ck.update(pin.p[i & 1], pc.len, i + 1 == pieces.size());
        he = hipSetDevice(pc.seg->device);
        if (he == hipSuccess) he = hipMemcpyAsync((uint8_t*)pc.seg->d_rows + pc.off, pin.p[i & 1], pc.len, hipMemcpyHostToDevice, pc.seg->stream);
    }
    if (pieces.empty()) ck.update(nullptr, 0, true);
    for (const Segment& sg : segs) {
        const hipError_t hs = hipStreamSynchronize(sg.stream);
        if (he == hipSuccess) he = hs;
    }
    close(fd);
    if (he != hipSuccess) return he == hipErrorOutOfMemory ? CQS_HIP_ERR_NOMEM : CQS_HIP_ERR_DEVICE;
    return ok && ck.finish() == checksum ? CQS_HIP_OK : CQS_HIP_ERR_INVALID;
}

}  // namespace cqs_idx

extern "C" {

int32_t cqs_hip_index_save(cqs_hip_index* x, const char* path, uint64_t* out_checksum) CQS_ABI_TRY {
    if (!x || !path) return CQS_HIP_ERR_INVALID;
    if (x->sh) return cqs_sharded::save(x, path, out_checksum);
    std::lock_guard<std::mutex> g(x->mu);
    if (x->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;  // src/cagra.rs:1103-1107
    HIP_TRY(x, hipSetDevice(x->device));
    HIP_TRY(x, quiesce(x));
    return save_segments(x, {Segment{x->device, x->d_rows, x->n, x->stream}}, x->dim, x->metric, path, out_checksum);
} CQS_ABI_CATCH(x)

int32_t cqs_hip_index_load(const char* path, uint32_t expected_dim, uint64_t expected_rows, int32_t device,
                           uint64_t row_base, cqs_hip_index** out) CQS_ABI_TRY {
    if (!path || !out) return CQS_HIP_ERR_INVALID;
    *out = nullptr;
    int fd = -1;
    uint64_t rows = 0, checksum = 0;
    uint32_t metric = 0;
    int32_t rc = open_blob(path, expected_dim, expected_rows, &fd, &rows, &metric, &checksum);
    if (rc != CQS_HIP_OK) return rc;
    cqs_hip_index* x = nullptr;
    rc = create_common(rows, expected_dim, metric, device, row_base, out, &x);
    if (rc != CQS_HIP_OK) { close(fd); return rc; }
    x->cap_rows = rows ? rows : 1;
    if (hipMalloc(&x->d_rows, (size_t)x->cap_rows * expected_dim * sizeof(float)) != hipSuccess) {
        close(fd);
        cqs_hip_index_destroy(x);
        return CQS_HIP_ERR_NOMEM;
    }
    rc = read_blob_into(fd, checksum, expected_dim, {Segment{device, x->d_rows, rows, x->stream}});
    if (rc != CQS_HIP_OK) { cqs_hip_index_destroy(x); return rc; }
    *out = x;
    return CQS_HIP_OK;
} CQS_ABI_CATCH_NOHANDLE

void cqs_hip_index_destroy(cqs_hip_index* x) CQS_ABI_TRY {
    if (!x) return;
    if (x->sh) { cqs_sharded::destroy(x); return; }
    hipSetDevice(x->device);
    (void)quiesce(x);  // src/cagra.rs:289-302 (incl. searches enqueued on caller streams)
    free_scratch(x);
    hipFree(x->d_keep);
    hipFree(x->d_dbg);
    if (!x->borrow) hipFree(x->d_rows);
    for (hipEvent_t e : x->ev) hipEventDestroy(e);
    if (x->done) hipEventDestroy(x->done);
    if (x->stream) hipStreamDestroy(x->stream);
    delete x;
} CQS_ABI_CATCH_VOID

uint64_t cqs_hip_index_len(const cqs_hip_index* x) CQS_ABI_TRY { return x ? (x->sh ? cqs_sharded::len(x) : x->n) : 0; } CQS_ABI_CATCH_VAL(0)
uint32_t cqs_hip_index_dim(const cqs_hip_index* x) CQS_ABI_TRY { return x ? x->dim : 0; } CQS_ABI_CATCH_VAL(0)
uint32_t cqs_hip_index_metric(const cqs_hip_index* x) CQS_ABI_TRY { return x ? x->metric : 0; } CQS_ABI_CATCH_VAL(0)
uint32_t cqs_hip_index_max_k(const cqs_hip_index* x) CQS_ABI_TRY { (void)x; return kMaxK; } CQS_ABI_CATCH_VAL(0)
int32_t cqs_hip_index_poisoned(const cqs_hip_index* x) CQS_ABI_TRY {
    if (x && x->sh) return cqs_sharded::poisoned(x);
    return x && x->poisoned.load(std::memory_order_acquire) ? 1 : 0;
} CQS_ABI_CATCH_NOHANDLE
int32_t cqs_hip_index_device(const cqs_hip_index* x) CQS_ABI_TRY { return x ? x->device : -1; } CQS_ABI_CATCH_NOHANDLE
uint64_t cqs_hip_index_row_base(const cqs_hip_index* x) CQS_ABI_TRY { return x ? x->row_base : 0; } CQS_ABI_CATCH_VAL(0)

size_t cqs_hip_index_last_error(const cqs_hip_index* x, char* buf, size_t cap) CQS_ABI_TRY {
    if (!x || !buf || cap == 0) return 0;
    if (x->sh) return cqs_sharded::last_error(x, buf, cap);
    std::lock_guard<std::mutex> g(x->mu);
    size_t m = x->last_error.size() < cap - 1 ? x->last_error.size() : cap - 1;
    memcpy(buf, x->last_error.data(), m);
    buf[m] = 0;
    return m;
} CQS_ABI_CATCH_VAL(0)

void cqs_hip_unpack_keys(const uint64_t* keys, size_t count, uint64_t* rows, float* scores) CQS_ABI_TRY {
    for (size_t i = 0; i < count; ++i) {
        const uint32_t ok = (uint32_t)(keys[i] >> 32);
        const uint32_t bits = (ok & 0x80000000u) ? (ok ^ 0x80000000u) : ~ok;
        float f;
        memcpy(&f, &bits, 4);
        if (scores) scores[i] = f;
        if (rows) rows[i] = (uint64_t)(0xFFFFFFFFu - (uint32_t)keys[i]);
    }
} CQS_ABI_CATCH_VOID

size_t cqs_hip_merge_keys(const uint64_t* lists, const uint32_t* counts, size_t n_lists, size_t stride, size_t k,
                          uint64_t* out_keys) CQS_ABI_TRY {
    // k-way merge of descending lists; n_lists is small (<= #GPUs), so a
    // linear scan over the list heads is cheaper than a heap.
    std::vector<size_t> pos(n_lists, 0);
    size_t outc = 0;
    while (outc < k) {
        size_t best = n_lists;
        uint64_t bk = 0;
        for (size_t l = 0; l < n_lists; ++l) {
            if (pos[l] < counts[l]) {
                const uint64_t v = lists[l * stride + pos[l]];
                if (best == n_lists || v > bk) { best = l; bk = v; }
            }
        }
        if (best == n_lists) break;
        out_keys[outc++] = bk;
        pos[best]++;
    }
    return outc;
} CQS_ABI_CATCH_VAL(0)

int32_t cqs_hip_index_search_device(cqs_hip_index* x, const float* d_queries, uint32_t b, uint32_t k,
                                    const uint32_t* d_keep_bitset, uint32_t mode, float threshold,
                                    uint64_t* d_out_keys, uint32_t* d_out_counts, void* stream) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_index_search_device");
    if (!x) return CQS_HIP_ERR_INVALID;
    if (x->sh) return CQS_HIP_ERR_INVALID;   // a row-sharded handle spans devices: host-buffer API only
    std::lock_guard<std::mutex> g(x->mu);
    if (x->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    if (b == 0) return CQS_HIP_OK;
    if (!d_queries || !d_out_keys || !d_out_counts) return fail(x, CQS_HIP_ERR_INVALID, "search_device: null buffer");
    if (k == 0 || k > kMaxK) return fail(x, CQS_HIP_ERR_INVALID, "search_device: k out of range");
    if (mode > CQS_HIP_MODE_PIPELINE) return fail(x, CQS_HIP_ERR_INVALID, "search_device: bad mode");
    if (b > max_query_block(x)) return fail(x, CQS_HIP_ERR_INVALID, "search_device: batch exceeds scratch budget");
    HIP_TRY(x, hipSetDevice(x->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the HIP null (legacy default) stream, e.g. torch's default stream
    if (x->n == 0) {
        HIP_TRY(x, hipMemsetAsync(d_out_counts, 0, (size_t)b * sizeof(uint32_t), st));
        return CQS_HIP_OK;
    }
    int32_t rc = ensure_scratch(x, b, k);
    if (rc != CQS_HIP_OK) return rc;
    return enqueue_search(x, d_queries, b, k, d_keep_bitset, mode, threshold, d_out_keys, d_out_counts, st);
} CQS_ABI_CATCH(x)

}  // extern "C"

namespace cqs_idx {

// Debug print of the kernels' stamps (CQS_HIP_DEBUG_STAMPS=1).  Caller holds mu; the stream is idle.
static void print_debug_stamps(cqs_hip_index* x) {
    unsigned long long h[16];   // 100 MHz realtime counter: 10 ns ticks
    if (hipMemcpy(h, x->d_dbg, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
        fprintf(stderr, "[cqs_hip] select_finish us: zero %.2f hist %.2f decide %.2f groups %.2f scores %.2f sort %.2f emit %.2f | groups=%llu cand=%llu\n",
                0.0, (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0, (h[4] - h[3]) / 100.0,
                (h[5] - h[4]) / 100.0, (h[6] - h[5]) / 100.0, h[8], h[9]);
    // scan waves: spread of start and end times relative to the first wave's start
    std::vector<unsigned long long> w(2 * cqs::kDbgWaves);
    if (hipMemcpy(w.data(), x->d_dbg + 16, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    std::vector<double> st, en;
    unsigned long long t0 = ~0ull;
    for (uint32_t i = 0; i < cqs::kDbgWaves; ++i) if (w[2 * i] && w[2 * i] < t0) t0 = w[2 * i];
    for (uint32_t i = 0; i < cqs::kDbgWaves; ++i)
        if (w[2 * i] && w[2 * i + 1]) { st.push_back((w[2 * i] - t0) / 100.0); en.push_back((w[2 * i + 1] - t0) / 100.0); }
    if (!st.empty()) {
        std::sort(st.begin(), st.end());
        std::sort(en.begin(), en.end());
        auto pc = [](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
        fprintf(stderr, "[cqs_hip] scan waves=%zu start us p0 %.1f p50 %.1f p90 %.1f p100 %.1f | end us p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f | select ends %.1f\n",
                st.size(), pc(st, 0), pc(st, .5), pc(st, .9), pc(st, 1), pc(en, 0), pc(en, .1), pc(en, .5), pc(en, .9), pc(en, 1),
                (h[6] - t0) / 100.0);
    }
    (void)hipMemset(x->d_dbg + 16, 0, w.size() * sizeof(unsigned long long));
}

// One host query of a block and where its answer goes.
struct HostQuery {
    const float* q;          // [dim]
    uint64_t* out_rows;      // [k]
    float* out_scores;       // [k]
    uint32_t* out_count;
};

// The host-buffer search proper: `b` queries with one (k, mode, threshold, bitset), scanned in blocks the scratch
// budget allows.  Caller holds mu, has checked the arguments and zeroed the counts.  `gemv_only`: every block goes
// through the HBM-streaming passes of <= 8 queries, whose scores do not depend on how many queries share a pass (same
// per-lane FMA chain, same butterfly) - what the combining queue needs to hand each caller the bits it would have got alone.
static int32_t search_host_locked(cqs_hip_index* x, const HostQuery* qs, uint32_t b, uint32_t k, const uint32_t* keep_bitset,
                                  uint32_t mode, float threshold, bool gemv_only) {
    if (x->inject_fail.exchange(0, std::memory_order_acq_rel) != 0)
        return fail(x, CQS_HIP_ERR_DEVICE, "search: injected device failure (test hook)");
    if (x->n == 0 || k == 0) return CQS_HIP_OK;               // src/cagra.rs:445-447
    HIP_TRY(x, hipSetDevice(x->device));
    // a device-API search on a caller stream may still use the shared scratch this call is about to overwrite
    if (x->done_valid && x->done_stream != x->stream) HIP_TRY(x, hipStreamWaitEvent(x->stream, x->done, 0));
    // bitset: count kept rows on the host (src/cagra.rs:747-775)
    const uint32_t* d_keep = nullptr;
    uint32_t k_eff = k;
    if (keep_bitset) {
        const uint64_t words = (x->n + 31) / 32;
        uint64_t included = 0;
        for (uint64_t w = 0; w < words; ++w) {
            uint32_t v = keep_bitset[w];
            if (w == words - 1 && (x->n % 32)) v &= (1u << (x->n % 32)) - 1u;
            included += (uint64_t)__builtin_popcount(v);
        }
        if (included == 0) return CQS_HIP_OK;                   // src/cagra.rs:765-767
        if (included < x->n) {                                  // all-pass == unfiltered, :760-762
            if (included < k_eff) k_eff = (uint32_t)included;   // :775
            int32_t rck = stage_keep(x, keep_bitset, words);
            if (rck != CQS_HIP_OK) return rck;
            d_keep = x->d_keep;
        }
    }

    const uint32_t blk = max_query_block(x);
    std::vector<uint8_t> bad(b, 0);
    for (uint32_t done = 0; done < b;) {
        const uint32_t nb = (b - done) < blk ? (b - done) : blk;
        int32_t rc = ensure_scratch(x, nb, k_eff);
        if (rc != CQS_HIP_OK) return rc;
        // stage queries; a non-finite query yields an empty result (src/cagra.rs:464-470)
        for (uint32_t i = 0; i < nb; ++i) {
            const float* src = qs[done + i].q;
            float* dst = x->h_q + (size_t)i * x->dim;
            bool ok = true;
            for (uint32_t d = 0; d < x->dim; ++d) ok &= std::isfinite(src[d]);
            bad[done + i] = !ok;
            if (ok) memcpy(dst, src, (size_t)x->dim * sizeof(float));
            else memset(dst, 0, (size_t)x->dim * sizeof(float));
        }
        HIP_TRY(x, hipMemcpyAsync(x->d_q, x->h_q, (size_t)nb * x->dim * sizeof(float), hipMemcpyHostToDevice, x->stream));
        // Small blocks: the select kernel writes keys and counts straight into the pinned host buffers (device-visible
        // addresses): no copy calls behind the kernels, one wait.  Large blocks keep the device buffers + two copies
        // (hundreds of KB of scattered 8-byte stores over PCIe would cost more than the copies).
        const bool direct = x->h_out_keys_dev && x->h_out_counts_dev && (size_t)nb * k_eff <= kDirectOutKeys;
        rc = enqueue_search(x, x->d_q, nb, k_eff, d_keep, mode, threshold, direct ? x->h_out_keys_dev : x->d_out_keys,
                            direct ? x->h_out_counts_dev : x->d_out_counts, x->stream, gemv_only);
        if (rc != CQS_HIP_OK) return rc;
        if (!direct) {
            HIP_TRY(x, hipMemcpyAsync(x->h_out_keys, x->d_out_keys, (size_t)nb * k_eff * sizeof(uint64_t), hipMemcpyDeviceToHost, x->stream));
            HIP_TRY(x, hipMemcpyAsync(x->h_out_counts, x->d_out_counts, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, x->stream));
        }
        HIP_TRY(x, hipStreamSynchronize(x->stream));
        if (x->d_dbg) print_debug_stamps(x);
        for (uint32_t i = 0; i < nb; ++i) {
            const uint32_t qi = done + i;
            if (bad[qi]) continue;
            uint32_t c = x->h_out_counts[i];
            if (c > k_eff) c = k_eff;
            cqs_hip_unpack_keys(x->h_out_keys + (size_t)i * k_eff, c, qs[qi].out_rows, qs[qi].out_scores);
            *qs[qi].out_count = c;
        }
        done += nb;
    }
    return CQS_HIP_OK;
}

// ---- the combining queue --------------------------------------------------------------------------------------------
// Concurrent single-query callers of cqs_hip_index_search (the daemon's client threads, src/cli/watch/daemon.rs:273,
// on one Arc<dyn VectorIndex>) used to queue on the handle mutex for one 0.5 ms pass EACH, although one pass scans up to
// 8 queries for 0.50-0.54 ms.  Now a caller parks its query; whoever leads next takes the device, gathers
// the parked queries with the same (k, mode, threshold) and runs them as ONE block of gemv passes; every caller gets
// exactly the bits a lone call would have produced (search_host_locked, gemv_only).  Bitsets, multi-query blocks and
// sharded handles keep the serial path.
constexpr uint32_t kCombineCap = 32;     // queries per combined block (4 passes of 8)

static bool same_params(const cqs_combine_req* a, const cqs_combine_req* b) {
    return a->k == b->k && a->mode == b->mode && memcmp(&a->thr, &b->thr, sizeof(float)) == 0;
}
static uint32_t count_like_front(const cqs_hip_index* x) {
    uint32_t n = 0;
    for (const cqs_combine_req* r : x->pending) n += same_params(r, x->pending.front()) ? 1u : 0u;
    return n;
}

// One sealed block on a single-device handle: the device mutex is taken here, for the pass alone.
static int32_t combine_run_single(cqs_hip_index* x, cqs_combine_req* const* batch, uint32_t nb) {
    std::lock_guard<std::mutex> dev(x->mu);       // (other entry points - device API searches, extend, save - order with the pass here)
    if (x->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    HostQuery hq[kCombineCap];
    for (uint32_t i = 0; i < nb; ++i) hq[i] = HostQuery{batch[i]->q, batch[i]->out_rows, batch[i]->out_scores, batch[i]->out_count};
    // gemv passes only: each caller gets its lone call's bits.  CQS_HIP_COMBINE_BITS=relaxed (opt-in, read at create) lets a
    // block of >= 9 callers take the matrix-core kernel instead - 32 queries per corpus sweep instead of 8, scores in another
    // summation order (|delta| <= 2e-6 on unit vectors, inside the parity tolerance; the reference's GPU backend promises no
    // bit-reproducibility across calls either, src/cagra.rs:443-492)
    const bool gemv_only = !(x->combine_relaxed && nb >= cqs::kMfmaMinQueries);
    return search_host_locked(x, hq, nb, batch[0]->k, nullptr, batch[0]->mode, batch[0]->thr, gemv_only);
}

// Lead one pass.  `lk` holds cmu on entry and on exit; x->leader is set by the caller.
static void combine_lead(cqs_hip_index* x, std::unique_lock<std::mutex>& lk) {
    // Stragglers: if recent passes carried more callers than are parked now, their threads are on their way back (a
    // caller needs some tens of microseconds between getting its answer and asking again).  Waiting for them costs a
    // little once; scanning without them costs them a whole pass.  The window is anchored at the END OF THE PREVIOUS
    // PASS (round 5), not at this leader's arrival: a caller that comes alone combine_wait_us or more after a burst does
    // not wait at all (round 4: it paid the full wait once), and a lone caller never waits (expect is 1).  No device
    // mutex is held meanwhile (round 4 spun inside x->mu): there is one leader at a time, so the device is only ever
    // contended by the other entry points, and those must not queue behind a spin.
    const uint32_t target = x->expect < kCombineCap ? x->expect : kCombineCap;
    if (x->combine_wait_us && count_like_front(x) < target) {
        const auto t_end = x->last_pass_end + std::chrono::microseconds(x->combine_wait_us);
        while (count_like_front(x) < target && std::chrono::steady_clock::now() < t_end) {
            lk.unlock();
            for (int i = 0; i < 64; ++i) __builtin_ia32_pause();
            lk.lock();
        }
    }
    // seal the block: the oldest request and everything parked with its parameters, oldest first
    cqs_combine_req* batch[kCombineCap];
    uint32_t nb = 0, left_like = 0;
    {
        const cqs_combine_req head = *x->pending.front();
        std::deque<cqs_combine_req*> keep;
        for (cqs_combine_req* r : x->pending) {
            if (same_params(r, &head)) {
                if (nb < kCombineCap) { batch[nb++] = r; continue; }
                ++left_like;
            }
            keep.push_back(r);
        }
        x->pending.swap(keep);
        x->n_pending.store((uint32_t)x->pending.size(), std::memory_order_relaxed);
    }
    x->expect = nb + left_like;                    // what this pass saw (>= 1)
    lk.unlock();

    int32_t rc = CQS_HIP_OK;
    try {
        rc = x->sh ? cqs_sharded::search_combined(x, batch, nb) : combine_run_single(x, batch, nb);
    } catch (const std::bad_alloc&) {
        rc = fail(x, CQS_HIP_ERR_NOMEM, "search: out of host memory");
    } catch (...) {
        rc = fail(x, CQS_HIP_ERR_INVALID, "search: unexpected C++ exception");
    }
    x->stat_passes.fetch_add(1, std::memory_order_relaxed);
    x->stat_queries.fetch_add(nb, std::memory_order_relaxed);
    const bool poisoned = x->sh ? cqs_sharded::poisoned(x) != 0 : x->poisoned.load(std::memory_order_acquire);

    lk.lock();
    x->last_pass_end = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < nb; ++i) {
        // the call that met the failure reports it; whoever rode along on a handle that is now poisoned gets what
        // any later call gets (src/cagra.rs:486-490)
        batch[i]->rc = (rc != CQS_HIP_OK && i > 0 && poisoned) ? CQS_HIP_ERR_POISONED : rc;
        batch[i]->done = true;
    }
    if (poisoned) {                                // nobody stays parked on a dead handle
        for (cqs_combine_req* r : x->pending) { r->rc = CQS_HIP_ERR_POISONED; r->done = true; }
        x->pending.clear();
        x->n_pending.store(0, std::memory_order_relaxed);
    } else if (!x->pending.empty()) {
        // callers that arrived during this pass with the same parameters could have ridden along: tell the next leader
        uint32_t like = 0;
        for (const cqs_combine_req* r : x->pending) like += same_params(r, batch[0]) ? 1u : 0u;
        if (nb + like > x->expect) x->expect = nb + like;
    }
}

static int32_t combine_search(cqs_hip_index* x, cqs_combine_req& r) {
    std::unique_lock<std::mutex> lk(x->cmu);
    x->pending.push_back(&r);
    x->n_pending.store((uint32_t)x->pending.size(), std::memory_order_relaxed);
    while (!r.done) {
        if (!x->leader) {
            x->leader = true;
            struct Reset {                             // whatever happens in there, the next caller can lead
                cqs_hip_index* x; std::unique_lock<std::mutex>& lk;
                ~Reset() { if (!lk.owns_lock()) lk.lock(); x->leader = false; x->ccv.notify_all(); }
            } reset{x, lk};
            combine_lead(x, lk);
        } else {
            x->ccv.wait(lk);
        }
    }
    return r.rc;
}

}  // namespace cqs_idx

extern "C" {

int32_t cqs_hip_index_search(cqs_hip_index* x, const float* queries, uint32_t b, uint32_t query_dim, uint32_t k,
                             const uint32_t* keep_bitset, uint32_t mode, float threshold, uint64_t* out_rows,
                             float* out_scores, uint32_t* out_counts) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_index_search");
    if (!x) return CQS_HIP_ERR_INVALID;
    // One query, no filter, arguments in order: the combining queue (dim is immutable; everything else the locked
    // path would check is checked here or inside the pass).  Round 5: a row-sharded parent takes it too - what a
    // multi-GPU daemon binds - its block runs through every shard and the host merge (cqs_sharded::search_combined).
    if (x->combine && b == 1 && !keep_bitset && queries && out_counts && out_rows && out_scores && query_dim == x->dim &&
        k >= 1 && k <= kMaxK && mode <= CQS_HIP_MODE_PIPELINE) {
        if (x->sh ? cqs_sharded::poisoned(x) != 0 : x->poisoned.load(std::memory_order_acquire))
            return CQS_HIP_ERR_POISONED;                                               // src/cagra.rs:486-490
        out_counts[0] = 0;
        bool finite = true;
        for (uint32_t d = 0; d < query_dim; ++d) finite &= std::isfinite(queries[d]);
        if (!finite) return CQS_HIP_OK;                                                // src/cagra.rs:464-470
        cqs_combine_req r{queries, k, mode, threshold, out_rows, out_scores, out_counts};
        return combine_search(x, r);
    }
    if (x->sh) return cqs_sharded::search(x, queries, b, query_dim, k, keep_bitset, mode, threshold, out_rows, out_scores, out_counts);
    std::lock_guard<std::mutex> g(x->mu);
    if (x->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;  // src/cagra.rs:486-490
    if (b == 0) return CQS_HIP_OK;
    if (!queries || !out_counts) return fail(x, CQS_HIP_ERR_INVALID, "search: null buffer");
    for (uint32_t i = 0; i < b; ++i) out_counts[i] = 0;
    if (x->n == 0 || k == 0) return CQS_HIP_OK;               // src/cagra.rs:445-447
    if (query_dim != x->dim) {                                  // src/cagra.rs:449-456
        x->last_error = "search: query dimension mismatch (empty result)";
        return CQS_HIP_OK;
    }
    if (k > kMaxK) return fail(x, CQS_HIP_ERR_INVALID, "search: k > max_k");
    if (mode > CQS_HIP_MODE_PIPELINE) return fail(x, CQS_HIP_ERR_INVALID, "search: bad mode");
    if (!out_rows || !out_scores) return fail(x, CQS_HIP_ERR_INVALID, "search: null output buffer");
    std::vector<HostQuery> hq(b);
    for (uint32_t i = 0; i < b; ++i)
        hq[i] = HostQuery{queries + (size_t)i * x->dim, out_rows + (size_t)i * k, out_scores + (size_t)i * k, out_counts + i};
    return search_host_locked(x, hq.data(), b, k, keep_bitset, mode, threshold, /*gemv_only=*/false);
} CQS_ABI_CATCH(x)

// Combining-queue counters since the handle was made: passes run by the queue and the queries they carried (bench /
// tests; not in the Rust trait).  Either pointer may be NULL.
void cqs_hip_index_combine_stats(const cqs_hip_index* x, uint64_t* passes, uint64_t* queries) CQS_ABI_TRY {
    if (passes) *passes = x ? x->stat_passes.load(std::memory_order_relaxed) : 0;
    if (queries) *queries = x ? x->stat_queries.load(std::memory_order_relaxed) : 0;
} CQS_ABI_CATCH_VOID

// `find_neighbors` (src/cli/commands/search/neighbors.rs:86-132) for a row of this index: the query is the
// target row where it already lies in HBM (no H2D), the scan asks for limit + 1 and the target itself is
// dropped from the answer: top-(limit+1) of all rows minus the target = top-limit of all rows but the target
// under the same total order (score desc, row asc; neighbors.rs:131), duplicates of the target included.
int32_t cqs_hip_index_neighbors(cqs_hip_index* x, uint64_t target_row, uint32_t limit, uint64_t* out_rows,
                                float* out_scores, uint32_t* out_count) CQS_ABI_TRY {
    CQS_ROCTX_RANGE("cqs_hip_index_neighbors");
    if (!x || !out_count) return CQS_HIP_ERR_INVALID;
    if (x->sh) return cqs_sharded::neighbors(x, target_row, limit, out_rows, out_scores, out_count);
    std::lock_guard<std::mutex> g(x->mu);
    *out_count = 0;
    if (x->poisoned.load(std::memory_order_acquire)) return CQS_HIP_ERR_POISONED;
    if (!out_rows || !out_scores) return fail(x, CQS_HIP_ERR_INVALID, "neighbors: null output buffer");
    if (target_row < x->row_base || target_row - x->row_base >= x->n)
        return fail(x, CQS_HIP_ERR_INVALID, "neighbors: target row not in this index");   // get_chunk_with_embedding fails, :98-106
    if (limit < 1u) limit = 1u;                      // limit.clamp(1, SIMILAR_LIMIT_MAX), neighbors.rs:95, cli/limits.rs:40
    if (limit > CQS_HIP_NEIGHBORS_MAX) limit = CQS_HIP_NEIGHBORS_MAX;
    if (x->n <= 1) return CQS_HIP_OK;
    const uint32_t k = (uint64_t)limit + 1u < x->n ? limit + 1u : (uint32_t)x->n;
    HIP_TRY(x, hipSetDevice(x->device));
    if (x->done_valid && x->done_stream != x->stream) HIP_TRY(x, hipStreamWaitEvent(x->stream, x->done, 0));
    int32_t rc = ensure_scratch(x, 1, k);
    if (rc != CQS_HIP_OK) return rc;
    const float* d_target = x->d_rows + (size_t)(target_row - x->row_base) * x->dim;
    rc = enqueue_search(x, d_target, 1, k, nullptr, CQS_HIP_MODE_RAW, 0.f, x->d_out_keys, x->d_out_counts, x->stream);
    if (rc != CQS_HIP_OK) return rc;
    HIP_TRY(x, hipMemcpyAsync(x->h_out_keys, x->d_out_keys, (size_t)k * sizeof(uint64_t), hipMemcpyDeviceToHost, x->stream));
    HIP_TRY(x, hipMemcpyAsync(x->h_out_counts, x->d_out_counts, sizeof(uint32_t), hipMemcpyDeviceToHost, x->stream));
    HIP_TRY(x, hipStreamSynchronize(x->stream));
    uint32_t c = x->h_out_counts[0] < k ? x->h_out_counts[0] : k, outc = 0;
    for (uint32_t i = 0; i < c && outc < limit; ++i) {
        uint64_t row;
        float score;
        cqs_hip_unpack_keys(x->h_out_keys + i, 1, &row, &score);
        if (row == target_row) continue;             // neighbors.rs:116-118 (exclude self)
        out_rows[outc] = row;
        out_scores[outc] = score;
        ++outc;
    }
    *out_count = outc;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(x)

void cqs_hip_index_set_timing(cqs_hip_index* x, int32_t enable) CQS_ABI_TRY {
    if (!x) return;
    if (x->sh) { cqs_sharded::set_timing(x, enable); return; }
    std::lock_guard<std::mutex> g(x->mu);
    x->timing = enable != 0;
    x->ev_used = 0;
} CQS_ABI_CATCH_VOID

int32_t cqs_hip_index_scan_time(cqs_hip_index* x, uint32_t* launches, double* total_ms) CQS_ABI_TRY {
    if (!x || !launches || !total_ms) return CQS_HIP_ERR_INVALID;
    if (x->sh) return cqs_sharded::scan_time(x, launches, total_ms);
    std::lock_guard<std::mutex> g(x->mu);
    *launches = 0;
    *total_ms = 0.0;
    HIP_TRY(x, hipSetDevice(x->device));
    for (size_t i = 0; i + 1 < x->ev_used; i += 2) {
        HIP_TRY(x, hipEventSynchronize(x->ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(x, hipEventElapsedTime(&ms, x->ev[i], x->ev[i + 1]));
        *total_ms += ms;
        *launches += 1;
    }
    x->ev_used = 0;
    return CQS_HIP_OK;
} CQS_ABI_CATCH(x)


// Test hook (not part of the public header): the next host search on this handle fails as a device error would
// (the handle ends up poisoned) - how tests/test_threads_gpu.py reaches the combining queue's failure path.
void cqs_hip_debug_index_fail_next(cqs_hip_index* x) CQS_ABI_TRY {
    if (x) x->inject_fail.store(1, std::memory_order_release);
} CQS_ABI_CATCH_VOID

// Bench aid (not part of the public header): `n_threads` native threads, each calling the PUBLIC blocking entry point
// cqs_hip_index_search `per_thread` times with one query at a time (thread t asks queries t, t + n_threads, ... of the
// `n_queries` host rows, round and round) - what the reference's daemon does with one thread per client
// (src/cli/watch/daemon.rs:273), without a Python interpreter lock between the callers.  out_rows / out_scores /
// out_counts [n_queries, k] / [n_queries] receive each query's last answer.  Returns wall seconds, < 0 on a failed call.
double cqs_hip_debug_client_storm(cqs_hip_index* x, const float* queries, uint32_t n_queries, uint32_t dim, uint32_t k,
                                  uint32_t n_threads, uint32_t per_thread, uint64_t* out_rows, float* out_scores,
                                  uint32_t* out_counts) CQS_ABI_TRY {
    if (!x || !queries || !n_queries || !n_threads || !out_rows || !out_scores || !out_counts) return -1.0;
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
                const int32_t rc = cqs_hip_index_search(x, queries + (size_t)qi * dim, 1, dim, k, nullptr, CQS_HIP_MODE_RAW, 0.f,
                                                        out_rows + (size_t)qi * k, out_scores + (size_t)qi * k, out_counts + qi);
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
