/*
 * cqs_oracle.h — CPU restatement of the cqs semantic-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / the reported CPU baseline.  The
 * product path (cqs_amd/, libcqs_hip.so) never links or calls it.
 *
 * The reference (jamie8johnson/cqs v1.51.0) is Rust and cannot be compiled in
 * the authoring container (no cargo/rustc), so this file restates, function by
 * function, the reference algorithm for the path; each function cites the
 * reference file:line it follows.  Parity pinning: the restatement is checked
 * against every known-answer test the reference's own test-suite holds for
 * this path (tests/test_oracle_kat.py lists them with their file:line).
 *
 * Third-party arithmetic restated here (absent from /root/reference):
 *   simsimd 6.5.16 (Cargo.lock:4049) `f32::dot` — published algorithm of
 *   simsimd_dot_f32_haswell: 8 f32 lanes, one FMA per lane per 8 elements,
 *   tail lanes zero-filled, horizontal reduction widened to f64, result
 *   returned as f64 and narrowed to f32 by the caller (src/math.rs:16-22).
 *   The reference pins this boundary only up to f32 tolerance
 *   (src/math.rs:95-215), which is what the parity tests use (1e-5).
 */
#ifndef CQS_ORACLE_H
#define CQS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A1: dot products ---------------------------------------------------- */
/* simsimd-style 8-lane f32 FMA dot, f64 horizontal reduce (src/math.rs:15-16): the haswell kernel's
 * arithmetic on every host (AVX2+FMA body when the CPU has both features, else scalar fmaf;
 * bit-identical).  This is the parity oracle's dot. */
double cqs_oracle_dot_simsimd(const float* a, const float* b, size_t n);
/* What simsimd's run-time dispatch would run on THIS host: the AVX-512 (skylake) kernel where
 * available, else the haswell one.  Timing baseline only (dot_kind 3). */
double cqs_oracle_dot_simsimd_native(const float* a, const float* b, size_t n);
/* "scalar-fmaf" / "avx2+fma" / "avx512f": the body behind _simsimd (which=0) / _native (which=1). */
const char* cqs_oracle_dot_isa(int which);
/* Test hook: pin the scalar bodies (1) / re-detect (0). */
void cqs_oracle_dot_force_scalar(int on);
/* f64-accumulate fallback (src/math.rs:18-22). */
double cqs_oracle_dot_f64(const float* a, const float* b, size_t n);
/* strict left-to-right f32 multiply-then-add (neighbors.rs:82, hnsw/mod.rs:291). */
float cqs_oracle_dot_seq_f32(const float* a, const float* b, size_t n);

/* cosine_similarity (src/math.rs:11-28). Returns 1 = Some(*out), 0 = None. */
int cqs_oracle_cosine_similarity(const float* a, size_t na, const float* b, size_t nb, float* out);
/* full_cosine_similarity (src/math.rs:35-67). Returns 1 = Some, 0 = None. */
int cqs_oracle_full_cosine_similarity(const float* a, size_t na, const float* b, size_t nb, float* out);

/* ---- A17: normalize_l2 (src/embedder/pooling.rs:60-67), in place ---------- */
void cqs_oracle_normalize_l2(float* v, size_t n);

/* ---- A2: BLOB <-> f32 (src/store/helpers/embeddings.rs:14-57) ------------- */
/* Returns 0 on success, -1 = EmbeddingBlobMismatch (len != dim*4). */
int cqs_oracle_bytes_to_embedding(const uint8_t* bytes, size_t len, size_t dim, float* out);

/* ---- A3: default-filter scoring pipeline (candidate.rs:538-578) ----------- */
/* clamp(score,0,1) then ThresholdGate `>=` (candidate.rs:506-520,550).
 * Returns 1 = Some(*out), 0 = None. */
int cqs_oracle_apply_scoring_default(float embedding_score, float threshold, float* out);

/* ---- A4: BoundedScoreHeap (candidate.rs:162-330) --------------------------
 * Two id flavours: u64 ids (integer row ids of the synthetic corpora) and
 * byte-string ids (chunk ids, compared as UTF-8 bytes like Rust `String`). */
typedef struct cqs_oracle_heap cqs_oracle_heap;
cqs_oracle_heap* cqs_oracle_heap_new(size_t capacity);
void   cqs_oracle_heap_free(cqs_oracle_heap*);
int    cqs_oracle_heap_would_accept(const cqs_oracle_heap*, float score);   /* candidate.rs:246-279 */
void   cqs_oracle_heap_push_u64(cqs_oracle_heap*, uint64_t id, float score); /* candidate.rs:281-323 */
void   cqs_oracle_heap_push_str(cqs_oracle_heap*, const char* id, float score);
size_t cqs_oracle_heap_len(const cqs_oracle_heap*);
/* into_sorted_vec (candidate.rs:325-334): score desc (total_cmp), id asc.
 * Writes up to cap entries; returns count.  For string heaps ids_out receives
 * the index of the pushed string in push order (0-based). */
size_t cqs_oracle_heap_into_sorted(cqs_oracle_heap*, uint64_t* ids_out, float* scores_out, size_t cap);

/* ---- A5: brute-force scan (search/query.rs:348-510 minus SQLite) ----------
 * corpus rows [n*dim] in RAM, ids = row index.  Per row: cosine (A1) -> None
 * skip -> clamp/threshold (A3) -> heap push (A4); then into_sorted_vec.
 * dot_kind: 0 simsimd-style (haswell order), 1 f64 fallback, 2 sequential f32,
 *           3 simsimd as dispatched on this host (timing baseline). */
size_t cqs_oracle_brute_force(const float* rows, size_t n, size_t dim, const float* query, size_t qdim,
                              size_t limit, float threshold, int dot_kind,
                              uint64_t* ids_out, float* scores_out);

/* ---- A6: find_neighbors (neighbors.rs:86-132): sequential dot, skip self,
 * full sort (score desc, id asc), truncate to clamp(limit,1,100). ----------- */
size_t cqs_oracle_find_neighbors(const float* rows, size_t n, size_t dim, size_t target_row,
                                 size_t limit, uint64_t* ids_out, float* scores_out);

/* ---- A7/A9: exact VectorIndex::search contract --------------------------
 * Guards of cagra.rs:443-470 (empty / k==0 / dim mismatch / non-finite query
 * -> 0 results); raw dot score; non-finite scores dropped (cagra.rs:649-651);
 * optional keep-bitset (cagra.rs:747-757: bit i of word i/32 keeps row i;
 * all-pass == unfiltered, none -> 0, k = min(k, included) cagra.rs:771);
 * order (score desc total_cmp, row asc).  mode: 0 raw, 1 = rank on
 * clamp(score,0,1) with threshold gate (A3 semantics). */
size_t cqs_oracle_index_search(const float* rows, size_t n, size_t dim, const float* query, size_t qdim,
                               size_t k, const uint32_t* keep_bitset, int mode, float threshold,
                               int dot_kind, uint64_t* ids_out, float* scores_out);

/* ---- A9/A10 score conversions ------------------------------------------- */
float cqs_oracle_cagra_cosine_from_l2sq(float d);      /* cagra.rs:656-661: (1 - d/2).min(1.0) */
float cqs_oracle_dist_dot_clamped(const float* a, const float* b, size_t n); /* hnsw/mod.rs:287-299 */

/* ---- A10: prepare_index_data skip rule (hnsw/mod.rs:688-746) --------------
 * keep[i] = 1 unless row i is all-zero or has a non-finite component.
 * Returns kept count. */
size_t cqs_oracle_prepare_index_keep(const float* rows, size_t n, size_t dim, uint8_t* keep);

/* ---- (f)5: SpladeIndex - the sparse retrieval leg (src/splade/index.rs:177-290) ---------------
 * build (index.rs:191-212): chunks in input order, `postings: token -> [(chunk_index, weight)]` in push order
 * (chunk order, and a document's own order for a token it names twice); id_map[chunk_index] = id.
 * Input here: the documents' sparse vectors as a forward CSR (doc_off[n + 1], tokens, weights). */
typedef struct cqs_oracle_splade cqs_oracle_splade;
cqs_oracle_splade* cqs_oracle_splade_build(const uint64_t* doc_off, const uint32_t* tokens, const float* weights, uint64_t n);
void   cqs_oracle_splade_free(cqs_oracle_splade*);
size_t cqs_oracle_splade_len(const cqs_oracle_splade*);             /* index.rs:294-296 */
size_t cqs_oracle_splade_unique_tokens(const cqs_oracle_splade*);   /* index.rs:304-306 */
/* search_with_filter (index.rs:223-290): empty query / empty index -> 0 (:237-239); for each query term IN QUERY ORDER,
 * for each posting of its token in list order: skip chunks the filter rejects (:253-255), `*scores.entry(chunk)
 * .or_insert(0.0) += query_weight * doc_weight` (:256; f32 multiply, then f32 add - no fma); then every SCORED chunk
 * through BoundedScoreHeap::would_accept / push (:267-279; non-finite scores never enter, ties keep the smaller id) and
 * into_sorted_vec (:281).  keep: nullable, one byte per chunk (the predicate evaluated on its id).  Chunk ids: `ids`
 * (n C strings, compared as bytes) if non-NULL, else `id_rank` (n u32: the rank of the chunk's id among all ids) if
 * non-NULL, else the chunk index itself.  Writes min(k, scored) chunk indices + scores, best first; returns the count. */
size_t cqs_oracle_splade_search(const cqs_oracle_splade*, const uint32_t* q_tokens, const float* q_weights, size_t n_terms,
                                size_t k, const uint8_t* keep, const char* const* ids, const uint32_t* id_rank,
                                uint64_t* chunks_out, float* scores_out);
/* postings the search above reads for this query (sum of the posting-list lengths of its terms): the unit of the
 * sparse leg's algorithmic bytes (bench.py) */
uint64_t cqs_oracle_splade_touched(const cqs_oracle_splade*, const uint32_t* q_tokens, size_t n_terms);

/* ---- limits / batch sizing ------------------------------------------------ */
size_t cqs_oracle_dim_scaled_batch(size_t baseline, size_t dim, size_t min, size_t max); /* limits.rs:292-300 */
size_t cqs_oracle_candidate_count_for(size_t limit, size_t floor);                       /* limits.rs:315-320 */
size_t cqs_oracle_embed_batch_size(size_t dim, size_t max_seq_length);                   /* models.rs:789-817 */

/* ---- A17 poolers (pooling.rs:87-175); hidden [b,s,d], mask [b,s] i64 ------ */
void cqs_oracle_mean_pool(const float* hidden, const int64_t* mask, size_t b, size_t s, size_t d, float* out);
void cqs_oracle_cls_pool(const float* hidden, size_t b, size_t s, size_t d, float* out);
void cqs_oracle_last_token_pool(const float* hidden, const int64_t* mask, size_t b, size_t s, size_t d, float* out);

/* ---- multi-threaded CPU baseline helper (bench.py cpu_baseline) -----------
 * Same as cqs_oracle_brute_force but over `threads` contiguous row shards with
 * a final merge by the same comparator (the reference itself is single-
 * threaded per query, search/query.rs:362).  Uses pthreads. */
/* Baseline placement (bench.py cpu_baseline): pin worker t to cpus[t % count] (count 0 = let the OS place threads), and
 * copy src into untouched dst pages with the same per-worker row partition, so each worker scans node-local memory. */
void cqs_oracle_set_worker_cpus(const int* cpus, int count);
void cqs_oracle_first_touch_copy(float* dst, const float* src, size_t n, size_t dim, int threads);
size_t cqs_oracle_brute_force_mt(const float* rows, size_t n, size_t dim, const float* query,
                                 size_t limit, float threshold, int threads, int dot_kind,
                                 uint64_t* ids_out, float* scores_out);

#ifdef __cplusplus
}
#endif
#endif
