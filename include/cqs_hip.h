/*
 * cqs_hip.h — C ABI of libcqs_hip.so, the MI355X (gfx950) implementation of the
 * cqs semantic-search hot path: exact brute-force dot/cosine scan + top-k over
 * an HBM-resident [n, dim] f32 corpus, and (embed section) the embedding
 * forward.  This is the drop-in boundary: plain pointers and sizes, opaque
 * handles, int32 status codes, no exceptions, no callbacks, no torch types.
 *
 * The reference (jamie8johnson/cqs, Rust) has no FFI for this path today; the
 * seams it would bind this library behind are cited per entry point
 * (file:line relative to the cqs repo root, v1.51.0).  INTEGRATION.md shows
 * the Rust `extern "C"` block + `impl VectorIndex` a maintainer would add.
 *
 * Threading: every entry point is safe to call concurrently on one handle
 * (the handle serialises device work behind an internal mutex, like
 * `CagraIndex.gpu: Mutex<GpuState>`, src/cagra.rs:263).  Searches of one handle share
 * one device scratch: `cqs_hip_index_search_device` returns with its kernels still in
 * flight, and the handle orders the NEXT search (any entry point, any stream) behind
 * them with an event, so calls on different streams serialise on the device instead of
 * racing; extend / save / destroy wait for the last enqueued search first.  Device failures never
 * abort: they return a negative status, set the handle's poisoned flag
 * (src/index.rs:203-205, src/cagra.rs:472-489) and leave a message for
 * cqs_hip_index_last_error.
 */
#ifndef CQS_HIP_H
#define CQS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------- */
#define CQS_HIP_OK              0
#define CQS_HIP_ERR_INVALID    (-1)  /* bad argument (null, k > max_k, dim unsupported, ...) */
#define CQS_HIP_ERR_DEVICE     (-2)  /* HIP runtime error; handle is poisoned */
#define CQS_HIP_ERR_NOMEM      (-3)  /* device or host allocation failed */
#define CQS_HIP_ERR_POISONED   (-4)  /* handle was poisoned by an earlier device error */
#define CQS_HIP_ERR_NO_DEVICE  (-5)  /* no usable gfx950 device */

/* DistanceMetric (src/index.rs:45-56).  Both rank by the raw inner product;
 * COSINE additionally promises unit-norm rows so that score == cosine and
 * `index_scores_are_cosine()` (src/index.rs:236-238) may return true. */
#define CQS_HIP_METRIC_COSINE 0u
#define CQS_HIP_METRIC_DOT    1u

/* Score mode of a search.
 * RAW     : rank and return the raw dot product — the `VectorIndex::search`
 *           contract (src/index.rs:139-146; CAGRA's score at src/cagra.rs:656-662).
 * PIPELINE: rank on clamp(score,0,1) and drop scores < threshold — what the
 *           reference's brute-force scan ranks on with a default SearchFilter
 *           (src/search/scoring/candidate.rs:550 clamp, :513-519 ThresholdGate,
 *           loop at src/search/query.rs:469-481). */
#define CQS_HIP_MODE_RAW      0u
#define CQS_HIP_MODE_PIPELINE 1u

/* Largest k one search call serves (`VectorIndex::max_k`, src/index.rs:219-221;
 * production asks for candidate_count_for(limit) = max(5*limit, 500),
 * src/limits.rs:315-320). */
#define CQS_HIP_MAX_K 1024u

typedef struct cqs_hip_index cqs_hip_index;

/* ---- library / device ------------------------------------------------------ */
const char* cqs_hip_version(void);
/* Number of visible HIP devices (0 if none / runtime unusable).  Mirrors
 * `CagraIndex::gpu_available` (src/cagra.rs:336-376) for backend selection. */
int32_t cqs_hip_device_count(void);
/* Free + total HBM bytes of `device` (gpu_available_for sizing, src/cagra.rs:336-376). */
int32_t cqs_hip_device_mem(int32_t device, uint64_t* free_bytes, uint64_t* total_bytes);

/* ---- index lifetime -------------------------------------------------------- */
/* Build an exact index over `n` rows of `dim` f32 (host, row-major, rows in
 * rowid order = id_map order; replaces `CagraIndex::build_from_flat`,
 * src/cagra.rs:922-960).  The rows are copied to HBM.  Nothing is skipped:
 * the caller pre-filters zero / non-finite rows exactly like
 * `prepare_index_data` (src/hnsw/mod.rs:688-746) so len() and row->id agree.
 * `row_base` is added to every emitted row id (0 for a whole corpus; the shard
 * offset when the corpus is row-sharded over several GPUs/processes).
 * dim must be a multiple of 4 and <= 4096 (the reference's presets reach 4096, src/embedder/models.rs:572). */
int32_t cqs_hip_index_create(const float* rows, uint64_t n, uint32_t dim, uint32_t metric,
                             int32_t device, uint64_t row_base, cqs_hip_index** out);
/* Same, from rows already resident in `device`'s HBM.  borrow != 0: the index
 * uses the caller's buffer in place (caller keeps it alive and unmodified);
 * borrow == 0: the rows are copied device-to-device. */
int32_t cqs_hip_index_create_device(const void* d_rows, uint64_t n, uint32_t dim, uint32_t metric,
                                    int32_t device, uint64_t row_base, int32_t borrow,
                                    cqs_hip_index** out);
/* Row-sharded index over several GPUs of ONE process (north_star): the reference's daemon is
 * one process holding an `Arc<dyn VectorIndex>` (src/cli/batch/context.rs:157, GPU state behind one mutex
 * src/cagra.rs:263), so the multi-GPU path sits behind the SAME handle type.  The n rows are cut in rowid order
 * into n_devices near-equal contiguous shards (shard starts are multiples of 256 rows), shard s lives on
 * devices[s]; every other entry point accepts the returned handle: `cqs_hip_index_search` broadcasts the query
 * block to every device, scans the shards concurrently, moves the per-shard packed (score,row) candidates with
 * ONE RCCL all-gather over xGMI (ncclGroupStart/End, communicators from ncclCommInitAll) and merges them on the
 * host with cqs_hip_merge_keys - the result contract is exactly the single-device one.  RCCL is bound at run time
 * (dlopen "librccl.so.1"); n_devices == 1 needs none.  A device may be named more than once (test hook for
 * one-GPU boxes): such a list cannot form an RCCL clique and gathers by device-to-device copies instead.
 * `cqs_hip_index_search_device` is not available on a sharded handle (-> CQS_HIP_ERR_INVALID); extend appends to
 * the last shard that holds rows; save writes one blob (rows of all shards in order), which either loader reads. */
int32_t cqs_hip_index_create_sharded(const float* rows, uint64_t n, uint32_t dim, uint32_t metric,
                                     const int32_t* devices, uint32_t n_devices, uint64_t row_base,
                                     cqs_hip_index** out);
int32_t cqs_hip_index_load_sharded(const char* path, uint32_t expected_dim, uint64_t expected_rows,
                                   const int32_t* devices, uint32_t n_devices, uint64_t row_base, cqs_hip_index** out);
/* Number of shards (1 for a single-device handle) and, per shard: its device, first global row, row count, and
 * whether its gathers go through RCCL (1) or device-to-device copies (0). */
uint32_t cqs_hip_index_shards(const cqs_hip_index* idx);
int32_t cqs_hip_index_shard_info(const cqs_hip_index* idx, uint32_t shard, int32_t* device, uint64_t* first_row,
                                 uint64_t* rows, int32_t* gathers_with_rccl);

/* Append rows (host) to an owning index — incremental add, the contract the
 * tiered backend's extend() exposes (src/tiered.rs:1-43).  Not valid on a
 * borrowing index. */
int32_t cqs_hip_index_extend(cqs_hip_index* idx, const float* rows, uint64_t n_new);
/* Synchronises the index's streams, then frees everything (src/cagra.rs:289-302). */
void cqs_hip_index_destroy(cqs_hip_index* idx);

/* ---- persistence (the `index.cagra` + `.meta` pair of the reference, src/cagra.rs:973-1157,
 * 1174-1330) -----------------------------------------------------------------
 * save: writes `path` = 64-byte header {magic "CQSHIPF1", version, dim, metric, rows, checksum} +
 * the raw little-endian f32 rows, through `path.tmp` + rename (atomic like
 * save_blob_atomic_with_rollback, src/cagra.rs:1468-1592).  *out_checksum receives the 64-bit
 * content checksum for the caller's sidecar (the shim writes the CagraMeta-style JSON: magic,
 * version, dim, chunk_count, id_map, checksum, metric).  A poisoned index refuses to save.
 * load: validates magic / version / dim / rows (expected_rows 0 = any) / file size / checksum;
 * any mismatch -> CQS_HIP_ERR_INVALID and the caller deletes the files and rebuilds
 * (src/cagra.rs:1739-1750). */
int32_t cqs_hip_index_save(cqs_hip_index* idx, const char* path, uint64_t* out_checksum);
int32_t cqs_hip_index_load(const char* path, uint32_t expected_dim, uint64_t expected_rows, int32_t device,
                           uint64_t row_base, cqs_hip_index** out);

/* ---- index properties (VectorIndex, src/index.rs:139-239) ------------------ */
uint64_t cqs_hip_index_len(const cqs_hip_index* idx);       /* len()  :149 */
uint32_t cqs_hip_index_dim(const cqs_hip_index* idx);       /* dim()  :160 */
uint32_t cqs_hip_index_metric(const cqs_hip_index* idx);
uint32_t cqs_hip_index_max_k(const cqs_hip_index* idx);     /* max_k() :219 */
int32_t  cqs_hip_index_poisoned(const cqs_hip_index* idx);  /* is_poisoned() :203 */
int32_t  cqs_hip_index_device(const cqs_hip_index* idx);
uint64_t cqs_hip_index_row_base(const cqs_hip_index* idx);
/* Copies the last error / warning text (NUL-terminated, truncated to cap). */
size_t cqs_hip_index_last_error(const cqs_hip_index* idx, char* buf, size_t cap);

/* ---- search: host buffers --------------------------------------------------
 * `VectorIndex::search` / `search_with_filter` for a block of `b` queries
 * (src/index.rs:146,167; the CAGRA exemplar src/cagra.rs:443-672,727-820).
 *
 * queries   [b * query_dim] f32, host.
 * query_dim must equal dim(); a mismatch yields out_counts[*] = 0 and
 *           CQS_HIP_OK (reference: warn + empty Vec, src/cagra.rs:449-456).
 * k         0 -> counts 0; k > max_k -> CQS_HIP_ERR_INVALID (callers trim
 *           with cap_k_to_backend, src/search/query.rs:232-245).
 * keep_bitset nullable, host, ceil(len/32) words, bit (i%32) of word i/32 = keep
 *           local row i (src/cagra.rs:747-757).  All-pass == unfiltered, none ->
 *           counts 0, k is capped at the number of kept rows (src/cagra.rs:760-775).
 * mode/threshold: see CQS_HIP_MODE_*.
 * A query with a non-finite component yields count 0 (src/cagra.rs:464-470).
 *
 * Result per query q: out_rows[q*k .. q*k+out_counts[q]) (row_base + local row)
 * and out_scores likewise, ordered by (score desc under f32 total order, row
 * asc) — the reference's order (candidate.rs:327, neighbors.rs:131) with
 * integer ids.  Non-finite scores are never emitted (src/cagra.rs:649-651).
 * Slots past out_counts[q] are untouched.
 *
 * Concurrent callers (the daemon calls `search` from one thread per client on a shared
 * Arc<dyn VectorIndex>, src/cli/watch/daemon.rs:273; CAGRA serialises them behind
 * Mutex<GpuState>, src/cagra.rs:263): calls with b == 1 and no bitset that meet on one
 * handle are COMBINED - a caller that finds the device busy parks its query, and whoever
 * takes the device next scans every parked query with the same (k, mode, threshold) in one
 * pass over the corpus (up to 8 queries share the HBM stream; more run as consecutive
 * passes inside the same hold of the device).  Each caller receives exactly the bytes a
 * lone call would have produced.  A lone caller pays two uncontended mutex operations.
 * A handle made by cqs_hip_index_create_sharded / _load_sharded combines the same way
 * (round 5): the block goes to every shard at once, one gather, one host merge per query.
 * Calls with a bitset or b > 1 run one after the other as before.
 * CQS_HIP_COMBINE_BITS=relaxed (read at create; default: exact) is an opt-in throughput mode: a block of >= 9 callers may
 * run on the matrix cores - 32 queries per corpus sweep instead of 8 (16 callers: 25 k q/s against 14 k) - and its answers
 * are then within the parity tolerance of the lone call's (scores <= 2e-6 apart on unit vectors, same ids outside
 * near-ties), not its bits.
 * If a pass fails, the call that led it returns the device error and every parked caller
 * returns CQS_HIP_ERR_POISONED.  CQS_HIP_COMBINE=0 (read at create) turns the queue off. */
int32_t cqs_hip_index_search(cqs_hip_index* idx, const float* queries, uint32_t b, uint32_t query_dim,
                             uint32_t k, const uint32_t* keep_bitset, uint32_t mode, float threshold,
                             uint64_t* out_rows, float* out_scores, uint32_t* out_counts);

/* ---- search: device buffers, asynchronous ----------------------------------
 * Same computation with every buffer in the index's device memory space and
 * no host synchronisation: work is enqueued on `stream` (a hipStream_t; NULL =
 * the HIP null stream, which is what PyTorch's default stream is) and the call
 * returns; results are ordered after it on that stream only.  Used by the bench (inputs
 * resident in HBM) and by the sharded multi-GPU path, whose per-shard
 * candidates feed an RCCL all-gather straight from HBM.
 * d_queries [b*dim] f32 must hold finite values (the caller checks; the host
 * entry point above does it for host queries).  d_keep_bitset nullable.
 * d_out_keys [b*k] u64 receives the packed candidates
 *     key = (ordered_u32(score) << 32) | (0xFFFFFFFF - global_row)
 * sorted descending (i.e. score desc, row asc), unused slots = 0.
 * d_out_counts [b] u32.  cqs_hip_unpack_keys decodes keys on the host. */
int32_t cqs_hip_index_search_device(cqs_hip_index* idx, const float* d_queries, uint32_t b, uint32_t k,
                                    const uint32_t* d_keep_bitset, uint32_t mode, float threshold,
                                    uint64_t* d_out_keys, uint32_t* d_out_counts, void* stream);

/* ---- neighbours of a stored row ----------------------------------------------
 * `find_neighbors` (src/cli/commands/search/neighbors.rs:86-132): exact kNN of a row that is already in the
 * index, the row itself excluded (:116-118), ordered (score desc, row asc) (:131), limit clamped to
 * [1, SIMILAR_LIMIT_MAX = 100] (:95, src/cli/limits.rs:40).  target_row is a GLOBAL row id (row_base + local).
 * out_rows / out_scores hold at least min(limit, 100) slots; *out_count <= min(limit, 100, len - 1).
 * target_row outside the index -> CQS_HIP_ERR_INVALID (the reference fails to load the target's embedding,
 * :98-106).  Scores are raw dots (the reference sums f32 left to right: equal to 1e-6 on unit vectors).
 * Rows with a non-finite score are dropped as in every search of this library; an index built through
 * `prepare_index_data` holds none. */
#define CQS_HIP_NEIGHBORS_MAX 100u
int32_t cqs_hip_index_neighbors(cqs_hip_index* idx, uint64_t target_row, uint32_t limit, uint64_t* out_rows,
                                float* out_scores, uint32_t* out_count);

/* Host helpers for packed candidate keys (see above). */
void cqs_hip_unpack_keys(const uint64_t* keys, size_t count, uint64_t* rows, float* scores);
/* Final host-side k-way merge of per-shard candidate lists for one query
 * (north_star: "RCCL all-gather of per-shard (score,id) candidates before the
 * final host-side merge").  lists = n_lists blocks of `stride` keys, of which
 * the first counts[i] are valid and sorted descending.  Writes the top
 * min(k, total) keys, sorted descending; returns that count. */
size_t cqs_hip_merge_keys(const uint64_t* lists, const uint32_t* counts, size_t n_lists, size_t stride,
                          size_t k, uint64_t* out_keys);

/* Combining-queue counters of a handle since it was made: passes the queue ran and the
 * queries they carried (queries / passes = mean callers per pass).  Either pointer may be
 * NULL.  Diagnostic; not part of the VectorIndex trait. */
void cqs_hip_index_combine_stats(const cqs_hip_index* idx, uint64_t* passes, uint64_t* queries);

/* ---- profiling aid ---------------------------------------------------------
 * With timing enabled every search brackets its dominant scan kernel launch(es)
 * with a HIP event pair on the launch stream (up to 4096 pairs between reads).
 * cqs_hip_index_scan_time synchronises those events, returns how many searches
 * were bracketed and their summed scan time in milliseconds, and resets the
 * pool.  bench.py derives roofline.achieved from it. */
void    cqs_hip_index_set_timing(cqs_hip_index* idx, int32_t enable);
int32_t cqs_hip_index_scan_time(cqs_hip_index* idx, uint32_t* launches, double* total_ms);

/* ==== embed section ==========================================================
 * EmbeddingGemma-300m forward (Gemma3 text encoder, bidirectional, + mean pool + 2 dense):
 * replaces the ONNX Runtime session the reference runs inside `Embedder::embed_batch`
 * (src/embedder/core.rs:1091-1203: inputs `input_ids` / `attention_mask` i64 [B, L]
 * right-padded with pad_id 0 by `pad_2d_i64_from_encodings`, src/embedder/pooling.rs:40-57;
 * output `sentence_embedding` f32 [B, 768], L2-normalised afterwards by the caller,
 * core.rs:1196-1203).  Tokenisation, prefixes, batching by `embed_batch_size()` and the
 * query caches stay on the host exactly as in the reference.
 * Compute: bf16 operands on the matrix cores, f32 accumulation, f32 residual stream. */
typedef struct cqs_hip_embedder cqs_hip_embedder;

typedef struct cqs_hip_embed_config {
    uint32_t vocab_size;       /* 262144 */
    uint32_t hidden;           /* 768  (multiple of 256) */
    uint32_t layers;           /* 24 */
    uint32_t heads;            /* 3 */
    uint32_t kv_heads;         /* 1   (heads / kv_heads <= 4) */
    uint32_t head_dim;         /* 256 (only value supported) */
    uint32_t intermediate;     /* 1152 (multiple of 64) */
    uint32_t dense_hidden;     /* 3072 (multiple of 128) */
    uint32_t sliding_window;   /* 512: config value; the bidirectional mask is |q-k| < window/2 + 1 */
    uint32_t sliding_pattern;  /* 6: layer i is full attention iff (i+1) % pattern == 0 */
    uint32_t max_seq;          /* 2048 (src/embedder/models.rs:455-470) */
    float rms_eps;             /* 1e-6 */
    float rope_theta_global;   /* 1e6 */
    float rope_theta_local;    /* 1e4 */
    float query_pre_attn_scalar; /* 256 */
} cqs_hip_embed_config;

/* Fills *cfg with the EmbeddingGemma-300m geometry (`ModelConfig::embeddinggemma_300m`,
 * src/embedder/models.rs:455-470; dimensions from the public model card). */
void cqs_hip_embed_config_default(cqs_hip_embed_config* cfg);

/* Lifecycle: create (empty) -> set_tensor for every weight -> finalize -> embed* -> destroy.
 * Tensor names are the Hugging Face Gemma3TextModel names (`embed_tokens.weight`,
 * `layers.N.self_attn.q_proj.weight`, ..., `norm.weight`) plus `dense1.weight` [dense_hidden, hidden]
 * and `dense2.weight` [hidden, dense_hidden]; data is f32 row-major and is converted to bf16
 * (matrices) or kept f32 (norm weights) on the device. */
int32_t cqs_hip_embedder_create(const cqs_hip_embed_config* cfg, int32_t device, cqs_hip_embedder** out);
int32_t cqs_hip_embedder_set_tensor(cqs_hip_embedder* e, const char* name, const float* data, uint64_t count);
int32_t cqs_hip_embedder_finalize(cqs_hip_embedder* e);
/* Loader replacing `create_session(model_path, ..)` (src/embedder/provider.rs:349-447).  Reads what the reference's
 * local-model hook holds (CQS_ONNX_DIR, src/embedder/download.rs:12-41): `<dir>/onnx/model.onnx` (structured layout,
 * src/embedder/models.rs:455-457) or `<dir>/model.onnx` (flat layout), with the external-data sidecar
 * `model.onnx_data` beside it (download.rs:82) - the float initialisers are parsed straight from the protobuf wire
 * format (no ONNX Runtime, no protobuf library): named parameters keep their names, `Linear` weights folded into
 * transposed MatMul initialisers are resolved through the consuming node's module path, the two Dense layers by
 * shape.  Without an ONNX file, a Hugging Face checkpoint directory is read instead: `model.safetensors`
 * (+ `2_Dense/model.safetensors`, `3_Dense/model.safetensors`, tensor `linear.weight`).  F32 / BF16 / F16 tensors. */
int32_t cqs_hip_embedder_load_dir(const char* model_dir, const cqs_hip_embed_config* cfg, int32_t device,
                                  cqs_hip_embedder** out);
void cqs_hip_embedder_destroy(cqs_hip_embedder* e);

uint32_t cqs_hip_embedder_dim(const cqs_hip_embedder* e);        /* `embedding_dim()` core.rs:961 */
uint32_t cqs_hip_embedder_max_seq(const cqs_hip_embedder* e);
int32_t  cqs_hip_embedder_poisoned(const cqs_hip_embedder* e);
size_t   cqs_hip_embedder_last_error(const cqs_hip_embedder* e, char* buf, size_t cap);

/* The `session.run` replacement.  input_ids / attention_mask: i64 [batch, seq_len], host; every
 * mask row must be 1...10...0 (right padding).  out: f32 [batch, hidden], host, NOT normalised
 * (the reference normalises per row right after, core.rs:1196-1203).  A row whose mask is all
 * zero yields a zero vector (src/embedder/pooling.rs:113-119).  Errors: bad mask / token id out
 * of range -> CQS_HIP_ERR_INVALID (`EmbedderError::InferenceFailed` in the shim). */
int32_t cqs_hip_embed(cqs_hip_embedder* e, const int64_t* input_ids, const int64_t* attention_mask,
                      uint32_t batch, uint32_t seq_len, float* out);
/* Asynchronous form of the same call, for the index pipeline (the embed stage is its bottleneck when the cache is
 * cold, src/cli/pipeline/embedding.rs:226-421): submit validates and packs the batch into pinned staging, enqueues
 * tables H2D + forward + D2H on one of the engine's two execution contexts (a HIP stream + activation scratch each;
 * consecutive tickets alternate, so two batches' kernel chains interleave on the device) and returns a ticket
 * without waiting; collect waits for that ticket and copies its [batch, hidden] rows out.  Up to 3 tickets may be in
 * flight: while the device runs batches i and i+1 the host packs batch i+2.  A 4th submit without a collect ->
 * CQS_HIP_ERR_INVALID.  Tickets may be collected in any order (and may finish out of order); results do not depend
 * on what else is in flight.  A slot is released only by collecting its ticket: a caller that gives up on a batch
 * (an error in a LATER submit, a cancelled index run) must still collect every ticket it holds - `out` = NULL
 * abandons one (waits for it, releases the slot, drops the rows).
 * submit_ragged takes the batch without padding: `tokens` = the sequences' ids back to back (i32), lens[b] = length
 * of sequence b (0 allowed -> zero vector) - what a length-sorted scheduler holds anyway. */
int32_t cqs_hip_embed_submit(cqs_hip_embedder* e, const int64_t* input_ids, const int64_t* attention_mask,
                             uint32_t batch, uint32_t seq_len, uint64_t* ticket);
int32_t cqs_hip_embed_submit_ragged(cqs_hip_embedder* e, const int32_t* tokens, const uint32_t* lens, uint32_t batch,
                                    uint64_t* ticket);
int32_t cqs_hip_embed_collect(cqs_hip_embedder* e, uint64_t ticket, float* out);
/* `Embedder::warm()` (src/embedder/core.rs:933-957): pay the first-call cost of the search-time forward before the
 * first real query.  One sequence of up to 128 tokens (`embed_query`) runs a kernel chain of its own, replayed from a
 * hipGraph kept per (token count, execution context); a graph is otherwise built the first time its length is seen
 * (eager chain + capture + instantiate on that query's clock).  warm builds and replays the graphs of every length
 * 1..max_tokens (clamped to what the search-time path serves).  Call it with no ticket in flight
 * (CQS_HIP_ERR_INVALID otherwise); it takes about a second for max_tokens = 128. */
int32_t cqs_hip_embedder_warm(cqs_hip_embedder* e, uint32_t max_tokens);
/* Search-time chain counters since the engine was made: graphs captured, captures that FAILED (that length runs its
 * chain eagerly from then on; the cause is in cqs_hip_embedder_last_error), graph replays, eager chain runs.  Any
 * pointer may be NULL.  Diagnostic: lets a caller (and the bench) see which path a query really took. */
void cqs_hip_embedder_query_graph_stats(const cqs_hip_embedder* e, uint64_t* captured, uint64_t* failed,
                                        uint64_t* replays, uint64_t* eager);
/* `normalize_l2` (src/embedder/pooling.rs:60-67) applied to each row of a host [n, dim] matrix, in place: f32
 * left-to-right sum of squares, scale by 1/sqrt when > 0, zero rows stay zero.  Host code (no device work). */
void cqs_hip_normalize_l2_rows(float* rows, uint64_t n, uint32_t dim);
/* Diagnostic twin: final-norm hidden states f32 [batch, seq_len, hidden] (zeros at padded positions). */
int32_t cqs_hip_embed_hidden(cqs_hip_embedder* e, const int64_t* input_ids, const int64_t* attention_mask,
                             uint32_t batch, uint32_t seq_len, float* out_hidden);
/* Milliseconds between the start and the end of the forward of the most recently COLLECTED batch (HIP events on its
 * stream; with other tickets in flight that interval includes the time it shared the device with them). */
float cqs_hip_embedder_last_ms(const cqs_hip_embedder* e);

/* ---- BERT-family auxiliary models ---------------------------------------------------------
 * The two other ONNX models of the reference reuse `create_session` (src/embedder/provider.rs) and are BERT
 * encoders with a small head.  One engine type, two heads:
 *   CQS_HIP_BERT_HEAD_MLM         SPLADE sparse encoder (naver/splade-cocondenser-ensembledistil: BERT-base masked-LM);
 *                                 replaces the `session.run` of `SpladeEncoder::encode` / `encode_batch`
 *                                 (src/splade/mod.rs:595-760, :774-1075)
 *   CQS_HIP_BERT_HEAD_CLASSIFIER  cross-encoder reranker (cross-encoder/ms-marco-MiniLM-L-6-v2); replaces the
 *                                 `session.run` of `Reranker::compute_scores_opt` (src/reranker.rs:343-533)
 *   CQS_HIP_BERT_HEAD_NONE        the BERT-family EMBEDDER presets of the `Embedder` seam (e5-base, v9-200k: BERT-base;
 *                                 bge-large, bge-large-ft: BERT-large; src/embedder/models.rs:346-405): replaces
 *                                 `session.run` + `mean_pool` / `cls_pool` (src/embedder/pooling.rs:87-128)
 * The tokenizer stays on the host as in the reference; token ids come in packed (sequences back to back + lengths).
 * Weights are handed over by HF tensor name (set_tensor, f32, copied) and frozen by finalize.  Calls on one engine
 * are serialised by an internal mutex.  Geometry limits: hidden a multiple of 128 (<= 1024), head dim 32 or 64, and
 * hidden, 3 x hidden and intermediate each a multiple of 192, 256 or 320 (a GEMM tile width). */
typedef struct cqs_hip_bert cqs_hip_bert;
enum { CQS_HIP_BERT_HEAD_MLM = 0, CQS_HIP_BERT_HEAD_CLASSIFIER = 1, CQS_HIP_BERT_HEAD_NONE = 2 /* encoder only: the BERT-family embedders */ };
typedef struct cqs_hip_bert_config {
    uint32_t vocab_size, hidden, layers, heads, intermediate, max_pos, type_vocab;
    uint32_t num_labels;   /* classifier head: outputs per sequence (1..16) */
    uint32_t head;         /* CQS_HIP_BERT_HEAD_* */
    float ln_eps;
} cqs_hip_bert_config;
/* Presets: head = MLM -> BERT-base / vocab 30522 (src/splade/mod.rs:120-150); head = CLASSIFIER -> MiniLM-L6-H384, one
 * label (src/reranker.rs:7,35); head = NONE -> BERT-base (e5-base / v9-200k; for bge-large set hidden 1024, layers 24,
 * heads 16, intermediate 4096: src/embedder/models.rs:346-405). */
int32_t cqs_hip_bert_config_default(uint32_t head, cqs_hip_bert_config* out);
int32_t cqs_hip_bert_create(const cqs_hip_bert_config* cfg, int32_t device, cqs_hip_bert** out);
/* name: the Hugging Face tensor name, with or without the leading `bert.` (e.g.
 * `encoder.layer.3.attention.self.query.weight`, `cls.predictions.transform.dense.bias`, `cls.predictions.bias`,
 * `pooler.dense.weight`, `classifier.weight`); the MLM decoder is tied to `embeddings.word_embeddings.weight`. */
int32_t cqs_hip_bert_set_tensor(cqs_hip_bert* e, const char* name, const float* data, uint64_t count);
int32_t cqs_hip_bert_finalize(cqs_hip_bert* e);
/* create + weights from a model directory + finalize: `{dir}/onnx/model.onnx` or `{dir}/model.onnx` (+ external data;
 * the bundle layout of src/reranker.rs:548-556 and src/splade/mod.rs:433-460), else a Hugging Face checkpoint
 * (`{dir}/model.safetensors`).  cfg gives the geometry (config.json stays the host's business,
 * src/splade/mod.rs:125-150). */
int32_t cqs_hip_bert_load_dir(const char* model_dir, const cqs_hip_bert_config* cfg, int32_t device, cqs_hip_bert** out);
void    cqs_hip_bert_destroy(cqs_hip_bert* e);
/* SPLADE: tokens = the sequences' ids back to back, lens[b] = tokens of sequence b (0 allowed: an all-zero row).
 * out_dense [batch, vocab] f32 = ln(1 + max(0, max over the sequence's tokens of the masked-LM logits)): the
 * pre-pooled `sparse_vector` output form (src/splade/mod.rs:960-978); the caller keeps (id, weight) with
 * weight > threshold, ascending id (src/splade/mod.rs:1049-1062).  The maximum is folded with a strict `>` from -inf
 * (src/splade/mod.rs:1033-1043). */
int32_t cqs_hip_splade_encode(cqs_hip_bert* e, const int32_t* tokens, const uint32_t* lens, uint32_t batch, float* out_dense);
/* The same with that threshold filter done on the device (no [batch, vocab] transfer, no host pass over it): sequence
 * b's (id, weight) pairs with weight > threshold, ascending id, at out_ids / out_weights [b * cap ..]; out_counts[b] =
 * how many passed - above `cap` the row was cut off after its first `cap` entries and the caller takes that sequence
 * through cqs_hip_splade_encode instead (trained models keep 100-300 entries, src/splade/mod.rs:44). */
int32_t cqs_hip_splade_encode_sparse(cqs_hip_bert* e, const int32_t* tokens, const uint32_t* lens, uint32_t batch,
                                     float threshold, uint32_t cap, uint32_t* out_ids, float* out_weights, uint32_t* out_counts);
/* Ticket form of the same (the index pipeline's SPLADE stage, src/splade/mod.rs:774-1075 called per batch): submit packs
 * the batch into pinned staging and enqueues it on one of the engine's two execution contexts (consecutive tickets
 * alternate) without waiting; collect waits for that ticket.  Up to 3 tickets in flight; a ticket is released only by
 * collecting it (out_ids = NULL abandons it).  The blocking call above is submit + collect. */
int32_t cqs_hip_splade_submit_sparse(cqs_hip_bert* e, const int32_t* tokens, const uint32_t* lens, uint32_t batch,
                                     float threshold, uint32_t cap, uint64_t* ticket);
int32_t cqs_hip_splade_collect_sparse(cqs_hip_bert* e, uint64_t ticket, uint32_t* out_ids, float* out_weights,
                                      uint32_t* out_counts);
/* Reranker: (query, passage) pairs as ids + token type ids (NULL = all zero), packed like the above; every sequence
 * non-empty.  out_logits [batch, num_labels] f32; score = sigmoid(out_logits[b * num_labels]) (src/reranker.rs:516-518). */
int32_t cqs_hip_rerank_logits(cqs_hip_bert* e, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens,
                              uint32_t batch, float* out_logits);
/* Embedder presets (head = NONE): pooled sentence vectors f32 [batch, hidden], NOT normalised (the caller applies
 * `normalize_l2`, src/embedder/core.rs:1196-1203).  pooling: 0 = mean over the sequence's tokens (`PoolingStrategy::Mean`,
 * an empty sequence gives zeros), 1 = first token (`PoolingStrategy::Cls`).  type_ids NULL = all zero. */
int32_t cqs_hip_bert_embed(cqs_hip_bert* e, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens,
                           uint32_t batch, uint32_t pooling, float* out);
/* Ticket form (the index pipeline's embed stage with a BERT-family preset, src/cli/pipeline/embedding.rs:226-421):
 * as cqs_hip_embed_submit / _collect; out = NULL abandons the ticket. */
int32_t cqs_hip_bert_embed_submit(cqs_hip_bert* e, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens,
                                  uint32_t batch, uint32_t pooling, uint64_t* ticket);
int32_t cqs_hip_bert_embed_collect(cqs_hip_bert* e, uint64_t ticket, float* out);
/* Diagnostic: final encoder hidden states of the packed tokens, f32 [sum(lens), hidden]. */
int32_t cqs_hip_bert_hidden(cqs_hip_bert* e, const int32_t* tokens, const int32_t* type_ids, const uint32_t* lens,
                            uint32_t batch, float* out_hidden);
uint32_t cqs_hip_bert_vocab(const cqs_hip_bert* e);
int32_t  cqs_hip_bert_poisoned(const cqs_hip_bert* e);
size_t   cqs_hip_bert_last_error(cqs_hip_bert* e, char* buf, size_t cap);

/* ---- sparse index: the SPLADE retrieval leg (`SpladeIndex`, src/splade/index.rs:177-306) ---------------------------
 * The consumer of the sparse vectors cqs_hip_splade_encode_sparse produces: `search_hybrid_inner` asks it for
 * candidate_count_for(limit) >= 500 chunks per query next to the dense leg (src/search/query.rs:898-901) and fuses the
 * two lists itself (:909-1010, stays in Rust).  Replaces the in-memory `postings: HashMap<u32, Vec<(usize, f32)>>` +
 * `HashMap<usize, f32>` accumulation by an HBM-resident posting array and one accumulate launch + the dense index's exact
 * select.  Scores are BIT-IDENTICAL to the reference's: every chunk's sum is built as 0.0 + qw*dw + ... in query-term order,
 * then posting order, f32 multiply and f32 add (index.rs:248-258).
 *
 * create (`SpladeIndex::build`, index.rs:191-212): the n chunks' sparse vectors in chunk-index order as a forward CSR -
 * doc_off [n + 1] (doc_off[0] = 0), tokens / weights [doc_off[n]]; any u32 token id, repeated tokens inside a document
 * kept (each occurrence is its own posting, as in the reference).  id_rank: NULL, or n u32 - a permutation of 0..n-1,
 * id_rank[i] = how many chunk ids sort before chunk i's id (bytes; equal ids in chunk order): equal scores are then
 * ordered by id like `BoundedScoreHeap` (src/search/scoring/candidate.rs:299-334); NULL orders them by chunk index.
 * The one weight bit pattern 0xFFFFFFFF (a NaN) is refused (-> CQS_HIP_ERR_INVALID); every other f32 is accepted.
 * n < 2^32 - 1024. */
typedef struct cqs_hip_sparse_index cqs_hip_sparse_index;
int32_t cqs_hip_sparse_index_create(const uint64_t* doc_off, const uint32_t* tokens, const float* weights, uint64_t n,
                                    const uint32_t* id_rank, int32_t device, cqs_hip_sparse_index** out);
/* The same index from the reference's OWN in-memory form - `postings: HashMap<u32, Vec<(usize, f32)>>` (index.rs:177-187), e.g.
 * right after `SpladeIndex::load` read the persisted file (index.rs:677-1072): token_ids [n_tokens] distinct keys in any order,
 * list_off [n_tokens + 1], post_chunks / post_weights [list_off[n_tokens]] = every key's list in its stored order.  Postings that
 * name a chunk >= n are dropped (the search skips them, index.rs:252); a list need not be ascending (postings of ONE chunk keep
 * their order, all the sums depend on); a key given twice -> CQS_HIP_ERR_INVALID.  Searches give the same bits as an index
 * built by cqs_hip_sparse_index_create from the documents. */
int32_t cqs_hip_sparse_index_create_inverted(const uint32_t* token_ids, const uint64_t* list_off, const uint32_t* post_chunks,
                                             const float* post_weights, uint64_t n_tokens, uint64_t n, const uint32_t* id_rank,
                                             int32_t device, cqs_hip_sparse_index** out);
/* Persistence - the role of `SpladeIndex::save` / `load` / `load_or_build` (index.rs:346-1107): a restart skips the rebuild; the
 * file is tied to the store's `splade_generation` counter (bumped on every write to `sparse_vectors`).  OWN format (the
 * reference's blake3-checksummed `splade.index.bin` is neither read nor written): 64-byte header {magic "CQSHIPS1", version,
 * ranked, chunks, tokens, postings, generation, checksum} + token ids, list offsets, postings and the id order, written to
 * `<path>.tmp`, fsync'ed and renamed over `path`.  load: wrong magic / version / generation / chunk count (expected_chunks
 * 0 = any) / size / checksum / structure -> CQS_HIP_ERR_INVALID, *out stays NULL and the caller builds from the rows
 * (index.rs:1073-1107); a missing file likewise. */
int32_t cqs_hip_sparse_index_save(cqs_hip_sparse_index* idx, const char* path, uint64_t generation, uint64_t* out_checksum);
int32_t cqs_hip_sparse_index_load(const char* path, uint64_t expected_chunks, uint64_t generation, int32_t device,
                                  cqs_hip_sparse_index** out);
void     cqs_hip_sparse_index_destroy(cqs_hip_sparse_index* idx);
uint64_t cqs_hip_sparse_index_len(const cqs_hip_sparse_index* idx);            /* index.rs:294-296 */
uint64_t cqs_hip_sparse_index_unique_tokens(const cqs_hip_sparse_index* idx);  /* index.rs:304-306 */
uint64_t cqs_hip_sparse_index_postings(const cqs_hip_sparse_index* idx);
/* `search_with_filter` (index.rs:223-290; `search` = keep_bitset NULL, :214-216).  The query's (token, weight) terms in
 * the caller's order; keep_bitset: NULL or ceil(n / 32) host words, bit i of word i / 32 = the filter predicate on chunk i
 * (evaluated by the caller on the chunk's id, as `search_hybrid_inner` builds its predicate, src/search/query.rs:861-877).
 * Writes min(k, scored chunks) entries, best first: out_chunks = chunk indices (positions in the build order = id_map
 * indices), out_scores = the raw dot products.  Chunks no posting of the query reaches are not candidates (even at score
 * 0 others may be); non-finite scores are dropped (`would_accept`, candidate.rs:245-247); an empty query, an empty index or
 * k = 0 give *out_count = 0 (index.rs:237-239).  k <= CQS_HIP_MAX_K. */
int32_t cqs_hip_sparse_index_search(cqs_hip_sparse_index* idx, const uint32_t* q_tokens, const float* q_weights, uint32_t n_terms,
                                    uint32_t k, const uint32_t* keep_bitset, uint64_t* out_chunks, float* out_scores,
                                    uint32_t* out_count);
/* The same for a batch of b <= 64 queries in ONE pair of launches (evaluation runs, or a caller that gathers its clients'
 * queries): query q's terms are q_tokens / q_weights [q_off[q], q_off[q + 1]) (q_off [b + 1], ascending); one optional filter for
 * all; out_chunks / out_scores [b][k] (row q holds out_counts[q] entries), out_counts [b].  Every query's answer is the one
 * cqs_hip_sparse_index_search gives for it alone, bit for bit. */
int32_t cqs_hip_sparse_index_search_batch(cqs_hip_sparse_index* idx, const uint64_t* q_off, const uint32_t* q_tokens,
                                          const float* q_weights, uint32_t b, uint32_t k, const uint32_t* keep_bitset,
                                          uint64_t* out_chunks, float* out_scores, uint32_t* out_counts);
/* Concurrent callers: unfiltered single-query cqs_hip_sparse_index_search calls on one handle are combined into shared
 * batches like the dense index's (CQS_HIP_COMBINE=0 / CQS_HIP_COMBINE_WAIT_US, read at create); every caller gets the bits
 * its own call would have produced.  Counters since the handle was made: batches run through the queue and the queries they
 * carried.  Either pointer may be NULL.  Diagnostic. */
void cqs_hip_sparse_index_combine_stats(const cqs_hip_sparse_index* idx, uint64_t* passes, uint64_t* queries);
/* Profiling aid: device time of the last search's accumulate launch (HIP events on its stream) and the postings it read
 * (the sum of its terms' list lengths: 8 bytes each = the launch's algorithmic bytes, with 4 bytes per chunk of score row).
 * Searches are bracketed by the two events only from the first call with a non-NULL accumulate_ms on (round 5: a caller
 * that never asks does not pay for them); that first call reports 0 ms for the search before it. */
int32_t cqs_hip_sparse_index_last_search(const cqs_hip_sparse_index* idx, float* accumulate_ms, uint64_t* touched_postings);
int32_t cqs_hip_sparse_index_poisoned(const cqs_hip_sparse_index* idx);
size_t  cqs_hip_sparse_index_last_error(const cqs_hip_sparse_index* idx, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* CQS_HIP_H */
