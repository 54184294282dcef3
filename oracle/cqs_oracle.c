/*
 * cqs_oracle.c — CPU restatement of the cqs semantic-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cqs_oracle.h).  Build: `make -C oracle`
 * (gcc -O2 -ffp-contract=off: only the explicit fmaf() calls below fuse, so
 * the arithmetic order is exactly what each function documents).
 *
 * Every function cites the reference file:line (relative to the cqs repo
 * root, v1.51.0) whose behaviour it restates.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* pthread_setaffinity_np */
#endif
#include "cqs_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* f32::total_cmp (Rust std): IEEE-754 totalOrder on the bit pattern.        */
static inline int32_t total_key(float x) {
    int32_t i;
    memcpy(&i, &x, 4);
    i ^= (int32_t)(((uint32_t)(i >> 31)) >> 1);
    return i;
}
static inline int total_cmp(float a, float b) {
    int32_t ka = total_key(a), kb = total_key(b);
    return (ka > kb) - (ka < kb);
}
/* Rust f32::clamp(min,max): `if self < min {min} else if self > max {max} else {self}`
 * (NaN and -0.0 pass through unchanged). */
static inline float rust_clamp(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

/* ---- A1 ------------------------------------------------------------------ */
/* simsimd 6.5.16 simsimd_dot_f32_haswell (published algorithm): ab_vec (8 f32
 * lanes) = fmadd(a_vec, b_vec, ab_vec) per 8 elements; a partial tail is loaded
 * zero-filled; the 8 lanes are reduced after widening to f64: lanes i and i+4
 * are added first (low/high 128-bit halves), then the 4 f64 sums pairwise.
 * Called from src/math.rs:15-16.
 *
 * Two bodies with the SAME arithmetic (one fused multiply-add per lane per 8
 * elements, same reduction): a scalar one (fmaf) and an AVX2+FMA one
 * (_mm256_fmadd_ps).  Results are bit-identical; the body is chosen once from
 * the CPU's feature bits (NOT from its model: `target_clones("arch=haswell")`
 * resolves to the vector clone on Intel Haswell parts only). */
static double reduce8_f64(const float* acc) {
    double s[4];
    for (int l = 0; l < 4; ++l) s[l] = (double)acc[l] + (double)acc[l + 4];
    double lo = s[0] + s[2], hi = s[1] + s[3];
    return lo + hi;
}
static double dot_haswell_scalar(const float* a, const float* b, size_t n) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(a[i + l], b[i + l], acc[l]);
    if (i < n) {
        for (int l = 0; l < 8; ++l) {
            float av = (i + l < n) ? a[i + l] : 0.0f;
            float bv = (i + l < n) ? b[i + l] : 0.0f;
            acc[l] = fmaf(av, bv, acc[l]);
        }
    }
    return reduce8_f64(acc);
}
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2,fma")))
static double dot_haswell_avx2(const float* a, const float* b, size_t n) {
    __m256 ab = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) ab = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), ab);
    if (i < n) {
        float ta[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t l = 0; i + l < n; ++l) { ta[l] = a[i + l]; tb[l] = b[i + l]; }
        ab = _mm256_fmadd_ps(_mm256_loadu_ps(ta), _mm256_loadu_ps(tb), ab);
    }
    float acc[8];
    _mm256_storeu_ps(acc, ab);
    return reduce8_f64(acc);
}
/* simsimd_dot_f32_skylake (AVX-512F; what simsimd's run-time dispatch takes on AVX-512 hosts): one
 * 16-lane fmadd accumulator, masked zero-filled tail, f32 tree reduction (512 -> 256 -> 128 -> hadd x2).
 * Used only as the timed CPU baseline on such hosts (cqs_oracle_dot_simsimd_native); parity and KATs
 * stay on the haswell order above, which is the same on every host. */
__attribute__((target("avx512f")))
static double dot_skylake_avx512(const float* a, const float* b, size_t n) {
    __m512 ab = _mm512_setzero_ps();
    size_t i = 0;
    for (; i + 16 <= n; i += 16) ab = _mm512_fmadd_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i), ab);
    if (i < n) {
        const __mmask16 m = (__mmask16)((1u << (n - i)) - 1u);
        ab = _mm512_fmadd_ps(_mm512_maskz_loadu_ps(m, a + i), _mm512_maskz_loadu_ps(m, b + i), ab);
    }
    __m512 x = _mm512_add_ps(ab, _mm512_shuffle_f32x4(ab, ab, _MM_SHUFFLE(0, 0, 3, 2)));
    __m128 r = _mm512_castps512_ps128(_mm512_add_ps(x, _mm512_shuffle_f32x4(x, x, _MM_SHUFFLE(0, 0, 0, 1))));
    r = _mm_hadd_ps(r, r);
    return (double)_mm_cvtss_f32(_mm_hadd_ps(r, r));
}
#endif

typedef double (*dot_fn)(const float*, const float*, size_t);
static dot_fn g_dot_haswell = 0, g_dot_native = 0;
static const char* g_isa_haswell = "scalar-fmaf";
static const char* g_isa_native = "scalar-fmaf";
static int g_force_scalar = 0;
static void dot_resolve(void) {
    dot_fn h = dot_haswell_scalar, nat = dot_haswell_scalar;
    const char *ih = "scalar-fmaf", *in = "scalar-fmaf";
#if defined(__x86_64__)
    __builtin_cpu_init();
    if (!g_force_scalar && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) {
        h = nat = dot_haswell_avx2;
        ih = in = "avx2+fma";
    }
    if (!g_force_scalar && __builtin_cpu_supports("avx512f")) {
        nat = dot_skylake_avx512;
        in = "avx512f";
    }
#endif
    g_isa_haswell = ih; g_isa_native = in;
    g_dot_native = nat;
    __atomic_store_n(&g_dot_haswell, h, __ATOMIC_RELEASE);
}
double cqs_oracle_dot_simsimd(const float* a, const float* b, size_t n) {
    dot_fn f = __atomic_load_n(&g_dot_haswell, __ATOMIC_ACQUIRE);
    if (!f) { dot_resolve(); f = g_dot_haswell; }
    return f(a, b, n);
}
double cqs_oracle_dot_simsimd_native(const float* a, const float* b, size_t n) {
    if (!__atomic_load_n(&g_dot_haswell, __ATOMIC_ACQUIRE)) dot_resolve();
    return g_dot_native(a, b, n);
}
/* ISA of the body behind cqs_oracle_dot_simsimd (which = 0) / _native (which = 1) on this host. */
const char* cqs_oracle_dot_isa(int which) {
    if (!__atomic_load_n(&g_dot_haswell, __ATOMIC_ACQUIRE)) dot_resolve();
    return which ? g_isa_native : g_isa_haswell;
}
/* Test hook: 1 = pin the scalar body (bit-identity check of the two bodies), 0 = re-detect. */
void cqs_oracle_dot_force_scalar(int on) {
    g_force_scalar = on;
    dot_resolve();
}

/* src/math.rs:18-22: `.map(|(&x,&y)| (x as f64)*(y as f64)).sum::<f64>()`. */
double cqs_oracle_dot_f64(const float* a, const float* b, size_t n) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];
    return s;
}

/* neighbors.rs:82 `a.iter().zip(b).map(|(x,y)| x*y).sum()` and
 * hnsw/mod.rs:291: f32 product then f32 add, strictly left to right. */
float cqs_oracle_dot_seq_f32(const float* a, const float* b, size_t n) {
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = a[i] * b[i];
        s = s + p;
    }
    return s;
}

static double dot_by_kind(const float* a, const float* b, size_t n, int kind) {
    switch (kind) {
        case 1: return cqs_oracle_dot_f64(a, b, n);
        case 2: return (double)cqs_oracle_dot_seq_f32(a, b, n);
        case 3: return cqs_oracle_dot_simsimd_native(a, b, n);
        default: return cqs_oracle_dot_simsimd(a, b, n);
    }
}

/* src/math.rs:11-28 */
static int cosine_kind(const float* a, size_t na, const float* b, size_t nb, int kind, float* out) {
    if (na != nb || na == 0) return 0;            /* math.rs:12-14 */
    float score = (float)dot_by_kind(a, b, na, kind); /* math.rs:16-22 `as f32` */
    if (isfinite(score)) {                         /* math.rs:23-27 */
        *out = score;
        return 1;
    }
    return 0;
}
int cqs_oracle_cosine_similarity(const float* a, size_t na, const float* b, size_t nb, float* out) {
    return cosine_kind(a, na, b, nb, 0, out);
}

/* src/math.rs:35-67 */
int cqs_oracle_full_cosine_similarity(const float* a, size_t na, const float* b, size_t nb, float* out) {
    if (na != nb || na == 0) return 0;
    double dot = 0.0, norm_a = 0.0, norm_b = 0.0;
    for (size_t i = 0; i < na; ++i) {
        double xd = a[i], yd = b[i];
        dot += xd * yd;
        norm_a += xd * xd;
        norm_b += yd * yd;
    }
    double denom = sqrt(norm_a) * sqrt(norm_b);
    if (denom == 0.0) return 0;
    float r = (float)(dot / denom);
    if (!isfinite(r)) return 0;
    *out = r;
    return 1;
}

/* ---- A17 ----------------------------------------------------------------- */
/* src/embedder/pooling.rs:60-67: norm_sq = fold(0.0, acc + x*x) in f32;
 * if norm_sq > 0 { inv = 1.0/sqrt(norm_sq); x *= inv }. */
void cqs_oracle_normalize_l2(float* v, size_t n) {
    float norm_sq = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float p = v[i] * v[i];
        norm_sq = norm_sq + p;
    }
    if (norm_sq > 0.0f) {
        float inv_norm = 1.0f / sqrtf(norm_sq);
        for (size_t i = 0; i < n; ++i) v[i] *= inv_norm;
    }
}

/* ---- A2 ------------------------------------------------------------------ */
/* src/store/helpers/embeddings.rs:31-57: length must equal dim*4 else
 * EmbeddingBlobMismatch; bytes are native little-endian f32; NaN/Inf pass. */
int cqs_oracle_bytes_to_embedding(const uint8_t* bytes, size_t len, size_t dim, float* out) {
    if (len != dim * 4) return -1;
    memcpy(out, bytes, len);
    return 0;
}

/* ---- A3 ------------------------------------------------------------------ */
/* candidate.rs:550 `embedding_score.clamp(0.0,1.0)`; with the default
 * SearchFilter and no notes every signal but ThresholdGate is disabled or the
 * identity, and ThresholdGate is `current >= threshold` (candidate.rs:513-519). */
int cqs_oracle_apply_scoring_default(float embedding_score, float threshold, float* out) {
    float base = rust_clamp(embedding_score, 0.0f, 1.0f);
    if (base >= threshold) {
        *out = base;
        return 1;
    }
    return 0;
}

/* ---- A4 ------------------------------------------------------------------ */
typedef struct {
    float score;
    uint64_t id;     /* u64 id, or push index for string heaps */
    char* sid;       /* NULL for u64 heaps */
    size_t slen;
} heap_ent;

struct cqs_oracle_heap {
    heap_ent* e;
    size_t len, capacity, pushes;
};

static int id_cmp(const heap_ent* a, const heap_ent* b) {
    if (a->sid || b->sid) { /* Rust String Ord: lexicographic bytes */
        size_t m = a->slen < b->slen ? a->slen : b->slen;
        int c = m ? memcmp(a->sid, b->sid, m) : 0;
        if (c) return c < 0 ? -1 : 1;
        return (a->slen > b->slen) - (a->slen < b->slen);
    }
    return (a->id > b->id) - (a->id < b->id);
}
/* Key order of `(OrderedFloat, Reverse<id>)` (candidate.rs:163): score by
 * total_cmp, then id reversed.  "worse" = smaller key = min-heap top. */
static int key_cmp(const heap_ent* a, const heap_ent* b) {
    int c = total_cmp(a->score, b->score);
    if (c) return c;
    return -id_cmp(a, b);
}
static void sift_up(heap_ent* e, size_t i) {
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (key_cmp(&e[i], &e[p]) >= 0) break;
        heap_ent t = e[i]; e[i] = e[p]; e[p] = t;
        i = p;
    }
}
static void sift_down(heap_ent* e, size_t n, size_t i) {
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && key_cmp(&e[l], &e[m]) < 0) m = l;
        if (r < n && key_cmp(&e[r], &e[m]) < 0) m = r;
        if (m == i) break;
        heap_ent t = e[i]; e[i] = e[m]; e[m] = t;
        i = m;
    }
}

cqs_oracle_heap* cqs_oracle_heap_new(size_t capacity) {
    cqs_oracle_heap* h = (cqs_oracle_heap*)calloc(1, sizeof(*h));
    h->capacity = capacity;
    h->e = (heap_ent*)calloc(capacity + 1, sizeof(heap_ent)); /* candidate.rs:219 */
    return h;
}
void cqs_oracle_heap_free(cqs_oracle_heap* h) {
    if (!h) return;
    for (size_t i = 0; i < h->len; ++i) free(h->e[i].sid);
    free(h->e);
    free(h);
}
size_t cqs_oracle_heap_len(const cqs_oracle_heap* h) { return h->len; }

/* candidate.rs:246-279 */
int cqs_oracle_heap_would_accept(const cqs_oracle_heap* h, float score) {
    if (!isfinite(score)) return 0;
    if (h->capacity == 0) return 0;
    if (h->len < h->capacity) return 1;
    return !(total_cmp(score, h->e[0].score) < 0);
}

/* candidate.rs:281-323 */
static void heap_push_ent(cqs_oracle_heap* h, heap_ent in) {
    h->pushes++;
    if (!isfinite(in.score)) { free(in.sid); return; }     /* :282-285 */
    if (h->len < h->capacity) {                            /* :288-291 */
        h->e[h->len] = in;
        sift_up(h->e, h->len);
        h->len++;
        return;
    }
    if (h->len == 0) { free(in.sid); return; }             /* capacity 0: peek() is None */
    heap_ent* worst = &h->e[0];                            /* :299 */
    int c = total_cmp(in.score, worst->score);
    int better = c > 0 || (c == 0 && id_cmp(&in, worst) < 0); /* :311-315 */
    if (better) {                                          /* :316-319 */
        free(worst->sid);
        h->e[0] = in;
        sift_down(h->e, h->len, 0);
    } else {
        free(in.sid);
    }
}
void cqs_oracle_heap_push_u64(cqs_oracle_heap* h, uint64_t id, float score) {
    heap_ent in = {score, id, NULL, 0};
    heap_push_ent(h, in);
}
void cqs_oracle_heap_push_str(cqs_oracle_heap* h, const char* id, float score) {
    heap_ent in;
    in.score = score;
    in.id = h->pushes;
    in.slen = strlen(id);
    in.sid = (char*)malloc(in.slen + 1);
    memcpy(in.sid, id, in.slen + 1);
    heap_push_ent(h, in);
}

/* candidate.rs:325-334: sort_by(|a,b| b.1.total_cmp(&a.1).then(a.0.cmp(&b.0))) */
static int sorted_cmp(const void* pa, const void* pb) {
    const heap_ent* a = (const heap_ent*)pa;
    const heap_ent* b = (const heap_ent*)pb;
    int c = total_cmp(b->score, a->score);
    if (c) return c;
    return id_cmp(a, b);
}
size_t cqs_oracle_heap_into_sorted(cqs_oracle_heap* h, uint64_t* ids_out, float* scores_out, size_t cap) {
    qsort(h->e, h->len, sizeof(heap_ent), sorted_cmp);
    size_t n = h->len < cap ? h->len : cap;
    for (size_t i = 0; i < n; ++i) {
        ids_out[i] = h->e[i].id;
        scores_out[i] = h->e[i].score;
    }
    return n;
}

/* ---- A5 ------------------------------------------------------------------ */
/* search/query.rs:453-484: for each row (rowid order): embedding_slice ->
 * score_candidate (cosine -> pipeline) -> score_heap.push; then
 * into_sorted_vec.  check_query_dim (query.rs:263): a query whose length
 * differs from the store dim is an error -> here 0 results. */
size_t cqs_oracle_brute_force(const float* rows, size_t n, size_t dim, const float* query, size_t qdim,
                              size_t limit, float threshold, int dot_kind,
                              uint64_t* ids_out, float* scores_out) {
    if (qdim != dim) return 0;
    cqs_oracle_heap* h = cqs_oracle_heap_new(limit);
    for (size_t r = 0; r < n; ++r) {
        float base, score;
        if (!cosine_kind(query, qdim, rows + r * dim, dim, dot_kind, &base)) continue; /* query.rs:476 */
        if (!cqs_oracle_apply_scoring_default(base, threshold, &score)) continue;
        cqs_oracle_heap_push_u64(h, (uint64_t)r, score);                              /* query.rs:479 */
    }
    size_t c = cqs_oracle_heap_into_sorted(h, ids_out, scores_out, limit);
    cqs_oracle_heap_free(h);
    return c;
}

/* ---- A6 ------------------------------------------------------------------ */
typedef struct { float score; uint64_t id; } scored;
static int scored_cmp(const void* pa, const void* pb) {
    const scored* a = (const scored*)pa;
    const scored* b = (const scored*)pb;
    int c = total_cmp(b->score, a->score);
    if (c) return c;
    return (a->id > b->id) - (a->id < b->id);
}
/* neighbors.rs:86-132 (limit.clamp(1, SIMILAR_LIMIT_MAX=100), cli/limits.rs:40) */
size_t cqs_oracle_find_neighbors(const float* rows, size_t n, size_t dim, size_t target_row,
                                 size_t limit, uint64_t* ids_out, float* scores_out) {
    if (limit < 1) limit = 1;
    if (limit > 100) limit = 100;
    if (target_row >= n) return 0;
    scored* s = (scored*)malloc(sizeof(scored) * (n ? n : 1));
    size_t m = 0;
    const float* t = rows + target_row * dim;
    for (size_t r = 0; r < n; ++r) {
        if (r == target_row) continue;                      /* neighbors.rs:116-118 */
        s[m].score = cqs_oracle_dot_seq_f32(t, rows + r * dim, dim);
        s[m].id = r;
        ++m;
    }
    qsort(s, m, sizeof(scored), scored_cmp);               /* neighbors.rs:131 */
    if (m > limit) m = limit;                              /* neighbors.rs:132 */
    for (size_t i = 0; i < m; ++i) { ids_out[i] = s[i].id; scores_out[i] = s[i].score; }
    free(s);
    return m;
}

/* ---- A7/A9 --------------------------------------------------------------- */
size_t cqs_oracle_index_search(const float* rows, size_t n, size_t dim, const float* query, size_t qdim,
                               size_t k, const uint32_t* keep_bitset, int mode, float threshold,
                               int dot_kind, uint64_t* ids_out, float* scores_out) {
    if (n == 0 || k == 0) return 0;                        /* cagra.rs:445-447 */
    if (qdim != dim) return 0;                             /* cagra.rs:449-456 */
    for (size_t i = 0; i < qdim; ++i)
        if (!isfinite(query[i])) return 0;                 /* cagra.rs:464-470 */
    size_t included = n;
    if (keep_bitset) {                                     /* cagra.rs:747-757 */
        included = 0;
        for (size_t i = 0; i < n; ++i)
            if (keep_bitset[i / 32] & (1u << (i % 32))) ++included;
        if (included == n) keep_bitset = NULL;             /* cagra.rs:760-762 */
        if (included == 0) return 0;                       /* cagra.rs:765-767 */
        if (k > included) k = included;                    /* cagra.rs:775 */
    }
    scored* s = (scored*)malloc(sizeof(scored) * n);
    size_t m = 0;
    for (size_t r = 0; r < n; ++r) {
        if (keep_bitset && !(keep_bitset[r / 32] & (1u << (r % 32)))) continue;
        float sc = (float)dot_by_kind(query, rows + r * dim, dim, dot_kind);
        if (!isfinite(sc)) continue;                       /* cagra.rs:649-651 */
        if (mode == 1 && !cqs_oracle_apply_scoring_default(sc, threshold, &sc)) continue;
        s[m].score = sc;
        s[m].id = r;
        ++m;
    }
    qsort(s, m, sizeof(scored), scored_cmp);
    if (m > k) m = k;
    for (size_t i = 0; i < m; ++i) { ids_out[i] = s[i].id; scores_out[i] = s[i].score; }
    free(s);
    return m;
}

/* cagra.rs:656-661 */
float cqs_oracle_cagra_cosine_from_l2sq(float d) {
    float v = 1.0f - d / 2.0f;
    return v < 1.0f ? v : 1.0f; /* f32::min(1.0); NaN.min(1.0) = 1.0 */
}
/* hnsw/mod.rs:287-299: `1.0 - dot.min(1.0)` */
float cqs_oracle_dist_dot_clamped(const float* a, const float* b, size_t n) {
    float dot = cqs_oracle_dot_seq_f32(a, b, n);
    float m = dot < 1.0f ? dot : 1.0f; /* NaN.min(1.0) = 1.0 */
    return 1.0f - m;
}

/* hnsw/mod.rs:717-731 */
size_t cqs_oracle_prepare_index_keep(const float* rows, size_t n, size_t dim, uint8_t* keep) {
    size_t kept = 0;
    for (size_t r = 0; r < n; ++r) {
        const float* v = rows + r * dim;
        int any_nonzero = 0, any_nonfinite = 0;
        for (size_t i = 0; i < dim; ++i) {
            if (v[i] != 0.0f) any_nonzero = 1;   /* NaN != 0.0 is true */
            if (!isfinite(v[i])) any_nonfinite = 1;
        }
        keep[r] = (uint8_t)(any_nonzero && !any_nonfinite);
        kept += keep[r];
    }
    return kept;
}

/* ---- limits -------------------------------------------------------------- */
static size_t sat_mul(size_t a, size_t b) {
    if (a != 0 && b > SIZE_MAX / a) return SIZE_MAX;
    return a * b;
}
static size_t clamp_sz(size_t v, size_t lo, size_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* limits.rs:292-300 */
size_t cqs_oracle_dim_scaled_batch(size_t baseline, size_t dim, size_t min, size_t max) {
    if (dim == 0) return clamp_sz(baseline, min, max);
    size_t scaled = sat_mul(baseline, 1024) / dim;
    return clamp_sz(scaled, min, max);
}
/* limits.rs:315-320 (floor = CQS_SEARCH_CANDIDATE_FLOOR, default 500) */
size_t cqs_oracle_candidate_count_for(size_t limit, size_t floor) {
    size_t v = sat_mul(limit, 5);
    return v > floor ? v : floor;
}
/* models.rs:802-817 */
size_t cqs_oracle_embed_batch_size(size_t dim, size_t max_seq_length) {
    double d = (double)(dim > 1 ? dim : 1);
    double s = (double)(max_seq_length > 1 ? max_seq_length : 1);
    double dim_factor = 1024.0 / d;
    double seq_factor = 512.0 / s;
    if (seq_factor < 0.25) seq_factor = 0.25;
    double v = 64.0 * dim_factor * seq_factor;
    if (v < 1.0) v = 1.0;
    size_t scaled = (size_t)v;
    size_t p = 1;
    while (p < scaled) p <<= 1; /* next_power_of_two */
    return clamp_sz(p, 2, 256);
}

/* ---- poolers ------------------------------------------------------------- */
/* pooling.rs:87-121: masked sum over seq / mask count; zero mask -> zeros. */
void cqs_oracle_mean_pool(const float* hidden, const int64_t* mask, size_t b, size_t s, size_t d, float* out) {
    for (size_t i = 0; i < b; ++i) {
        float count = 0.0f;
        for (size_t j = 0; j < s; ++j) count += (float)mask[i * s + j];
        for (size_t c = 0; c < d; ++c) {
            float sum = 0.0f;
            for (size_t j = 0; j < s; ++j) sum += hidden[(i * s + j) * d + c] * (float)mask[i * s + j];
            out[i * d + c] = count > 0.0f ? sum / count : 0.0f;
        }
    }
}
/* pooling.rs:128-133 */
void cqs_oracle_cls_pool(const float* hidden, size_t b, size_t s, size_t d, float* out) {
    for (size_t i = 0; i < b; ++i) memcpy(out + i * d, hidden + i * s * d, d * sizeof(float));
}
/* pooling.rs:145-175: rightmost mask!=0 position, else 0 */
void cqs_oracle_last_token_pool(const float* hidden, const int64_t* mask, size_t b, size_t s, size_t d, float* out) {
    for (size_t i = 0; i < b; ++i) {
        size_t last = 0;
        for (size_t j = s; j-- > 0;)
            if (mask[i * s + j] != 0) { last = j; break; }
        memcpy(out + i * d, hidden + (i * s + last) * d, d * sizeof(float));
    }
}

/* ---- multi-threaded baseline --------------------------------------------- */
typedef struct {
    const float* rows; size_t lo, hi, dim; const float* query; size_t limit; float threshold; int dot_kind;
    uint64_t* ids; float* scores; size_t count; int t;
} mt_job;
/* Worker placement for the multi-thread baseline (bench.py's cpu_baseline): with pinning on, worker t of T runs on the
 * t-th CPU of the caller's list and stays there, so that a corpus whose shard t was FIRST TOUCHED by worker t
 * (cqs_oracle_first_touch_copy) is read from that worker's own NUMA node.  Off (the default): the OS places the threads. */
static int g_pin_count = 0;
static int g_pin_cpus[1024];
void cqs_oracle_set_worker_cpus(const int* cpus, int count) {
    if (!cpus || count < 0) count = 0;
    if (count > 1024) count = 1024;
    for (int i = 0; i < count; ++i) g_pin_cpus[i] = cpus[i];
    g_pin_count = count;
}
static void pin_worker(int t) {
    if (g_pin_count <= 0) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(g_pin_cpus[t % g_pin_count], &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}
typedef struct { float* dst; const float* src; size_t lo, hi, dim; int t; } ft_job;
static void* ft_worker(void* p) {
    ft_job* j = (ft_job*)p;
    pin_worker(j->t);
    if (j->hi > j->lo) memcpy(j->dst + j->lo * j->dim, j->src + j->lo * j->dim, (j->hi - j->lo) * j->dim * sizeof(float));
    return NULL;
}
/* dst (untouched pages, e.g. a fresh numpy.empty) <- src, shard t of `threads` copied by worker t: the same row
 * partition cqs_oracle_brute_force_mt uses, so each scan worker later reads pages its own node holds. */
void cqs_oracle_first_touch_copy(float* dst, const float* src, size_t n, size_t dim, int threads) {
    if (threads < 1) threads = 1;
    ft_job* jobs = (ft_job*)calloc((size_t)threads, sizeof(ft_job));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        size_t lo = per * (size_t)t, hi = lo + per;
        if (lo > n) lo = n;
        if (hi > n) hi = n;
        jobs[t] = (ft_job){dst, src, lo, hi, dim, t};
        pthread_create(&th[t], NULL, ft_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
}

static void* mt_worker(void* p) {
    mt_job* j = (mt_job*)p;
    pin_worker(j->t);
    cqs_oracle_heap* h = cqs_oracle_heap_new(j->limit);
    for (size_t r = j->lo; r < j->hi; ++r) {
        float base, score;
        if (!cosine_kind(j->query, j->dim, j->rows + r * j->dim, j->dim, j->dot_kind, &base)) continue;
        if (!cqs_oracle_apply_scoring_default(base, j->threshold, &score)) continue;
        cqs_oracle_heap_push_u64(h, (uint64_t)r, score);
    }
    j->count = cqs_oracle_heap_into_sorted(h, j->ids, j->scores, j->limit);
    cqs_oracle_heap_free(h);
    return NULL;
}
size_t cqs_oracle_brute_force_mt(const float* rows, size_t n, size_t dim, const float* query,
                                 size_t limit, float threshold, int threads, int dot_kind,
                                 uint64_t* ids_out, float* scores_out) {
    if (threads < 1) threads = 1;
    mt_job* jobs = (mt_job*)calloc((size_t)threads, sizeof(mt_job));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        size_t lo = per * (size_t)t, hi = lo + per;
        if (lo > n) lo = n;
        if (hi > n) hi = n;
        jobs[t] = (mt_job){rows, lo, hi, dim, query, limit, threshold, dot_kind,
                           (uint64_t*)malloc(sizeof(uint64_t) * (limit ? limit : 1)),
                           (float*)malloc(sizeof(float) * (limit ? limit : 1)), 0, t};
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    cqs_oracle_heap* h = cqs_oracle_heap_new(limit);
    for (int t = 0; t < threads; ++t) {
        pthread_join(th[t], NULL);
        for (size_t i = 0; i < jobs[t].count; ++i) cqs_oracle_heap_push_u64(h, jobs[t].ids[i], jobs[t].scores[i]);
        free(jobs[t].ids);
        free(jobs[t].scores);
    }
    size_t c = cqs_oracle_heap_into_sorted(h, ids_out, scores_out, limit);
    cqs_oracle_heap_free(h);
    free(jobs);
    free(th);
    return c;
}

/* ================= (f)5 SpladeIndex (src/splade/index.rs:177-290) ================= */
struct cqs_oracle_splade {
    uint64_t n;                 /* id_map.len() */
    size_t n_tokens;            /* postings.len() */
    uint32_t* tok;              /* sorted distinct token ids (the HashMap's keys) */
    uint64_t* off;              /* [n_tokens + 1] */
    uint32_t* p_chunk;          /* postings in push order per token */
    float* p_w;
};

static int cmp_u32(const void* a, const void* b) {
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static size_t splade_find(const cqs_oracle_splade* s, uint32_t token) {   /* postings.get(&token_id): index or (size_t)-1 */
    size_t a = 0, b = s->n_tokens;
    while (a < b) {
        const size_t m = (a + b) / 2;
        if (s->tok[m] < token) a = m + 1; else b = m;
    }
    return (a < s->n_tokens && s->tok[a] == token) ? a : (size_t)-1;
}

cqs_oracle_splade* cqs_oracle_splade_build(const uint64_t* doc_off, const uint32_t* tokens, const float* weights, uint64_t n) {
    cqs_oracle_splade* s = (cqs_oracle_splade*)calloc(1, sizeof *s);
    if (!s) return NULL;
    const uint64_t P = n ? doc_off[n] : 0;
    s->n = n;
    /* the HashMap's key set, sorted; slot lookup through a direct table when the ids are small (every real vocabulary),
     * else by bisection - which way the table is found does not change what it holds */
    uint32_t max_tok = 0;
    for (uint64_t i = 0; i < P; ++i) if (tokens[i] > max_tok) max_tok = tokens[i];
    uint32_t* direct = NULL;                                   /* token -> slot + 1 */
    if (P && max_tok < (1u << 24)) {
        direct = (uint32_t*)calloc((size_t)max_tok + 1, sizeof(uint32_t));
        for (uint64_t i = 0; i < P; ++i) direct[tokens[i]] = 1;
        size_t u = 0;
        for (uint32_t t = 0; t <= max_tok; ++t) if (direct[t]) u++;
        s->tok = (uint32_t*)malloc((u ? u : 1) * sizeof(uint32_t));
        u = 0;
        for (uint32_t t = 0; t <= max_tok; ++t) if (direct[t]) { s->tok[u] = t; direct[t] = (uint32_t)(++u); }
        s->n_tokens = u;
    } else {
        uint32_t* sorted = (uint32_t*)malloc((P ? P : 1) * sizeof(uint32_t));
        if (P) memcpy(sorted, tokens, P * sizeof(uint32_t));
        qsort(sorted, P, sizeof(uint32_t), cmp_u32);
        size_t u = 0;
        for (uint64_t i = 0; i < P; ++i)
            if (i == 0 || sorted[i] != sorted[i - 1]) sorted[u++] = sorted[i];
        s->n_tokens = u;
        s->tok = sorted;
    }
    const size_t u = s->n_tokens;
#define SPLADE_SLOT(tk) (direct ? (size_t)direct[(tk)] - 1 : splade_find(s, (tk)))
    s->off = (uint64_t*)calloc(u + 1, sizeof(uint64_t));
    s->p_chunk = (uint32_t*)malloc((P ? P : 1) * sizeof(uint32_t));
    s->p_w = (float*)malloc((P ? P : 1) * sizeof(float));
    for (uint64_t i = 0; i < P; ++i) s->off[SPLADE_SLOT(tokens[i]) + 1]++;
    for (size_t t = 0; t < u; ++t) s->off[t + 1] += s->off[t];
    uint64_t* cur = (uint64_t*)malloc((u ? u : 1) * sizeof(uint64_t));
    memcpy(cur, s->off, u * sizeof(uint64_t));
    for (uint64_t d = 0; d < n; ++d)                                     /* index.rs:197-202: chunk by chunk, entry by entry */
        for (uint64_t e = doc_off[d]; e < doc_off[d + 1]; ++e) {
            const uint64_t at = cur[SPLADE_SLOT(tokens[e])]++;
            s->p_chunk[at] = (uint32_t)d;
            s->p_w[at] = weights[e];
        }
#undef SPLADE_SLOT
    free(direct);
    free(cur);
    return s;
}
void cqs_oracle_splade_free(cqs_oracle_splade* s) {
    if (!s) return;
    free(s->tok); free(s->off); free(s->p_chunk); free(s->p_w); free(s);
}
size_t cqs_oracle_splade_len(const cqs_oracle_splade* s) { return (size_t)s->n; }
size_t cqs_oracle_splade_unique_tokens(const cqs_oracle_splade* s) { return s->n_tokens; }
uint64_t cqs_oracle_splade_touched(const cqs_oracle_splade* s, const uint32_t* q_tokens, size_t n_terms) {
    uint64_t t = 0;
    for (size_t i = 0; i < n_terms; ++i) {
        const size_t j = splade_find(s, q_tokens[i]);
        if (j != (size_t)-1) t += s->off[j + 1] - s->off[j];
    }
    return t;
}

size_t cqs_oracle_splade_search(const cqs_oracle_splade* s, const uint32_t* q_tokens, const float* q_weights, size_t n_terms,
                                size_t k, const uint8_t* keep, const char* const* ids, const uint32_t* id_rank,
                                uint64_t* chunks_out, float* scores_out) {
    if (n_terms == 0 || s->n == 0) return 0;                              /* index.rs:237-239 */
    /* `scores: HashMap<usize, f32>` as a dense array + a presence flag: same values, and the heap below does not
     * depend on the order the map is walked in (ids are distinct, its order is total) */
    float* score = (float*)calloc((size_t)s->n, sizeof(float));
    uint8_t* scored = (uint8_t*)calloc((size_t)s->n, 1);
    for (size_t i = 0; i < n_terms; ++i) {                                 /* :248 */
        const size_t j = splade_find(s, q_tokens[i]);
        if (j == (size_t)-1) continue;                                     /* :249 */
        const float qw = q_weights[i];
        for (uint64_t e = s->off[j]; e < s->off[j + 1]; ++e) {             /* :250 */
            const uint32_t c = s->p_chunk[e];
            if (c >= s->n || (keep && !keep[c])) continue;                 /* :252-254 */
            const float prod = qw * s->p_w[e];                             /* :256 (-ffp-contract=off: no fma) */
            if (!scored[c]) { scored[c] = 1; score[c] = 0.0f; }
            score[c] = score[c] + prod;
        }
    }
    cqs_oracle_heap* h = cqs_oracle_heap_new(k);                           /* :265 */
    /* string ids: into_sorted returns push order, so remember which chunk each push was */
    uint64_t* pushed = ids ? (uint64_t*)malloc((size_t)s->n * sizeof(uint64_t)) : NULL;
    size_t np = 0;
    for (uint64_t c = 0; c < s->n; ++c) {
        if (!scored[c]) continue;
        if (!cqs_oracle_heap_would_accept(h, score[c])) continue;          /* :270-272 */
        if (ids) { cqs_oracle_heap_push_str(h, ids[c], score[c]); pushed[np++] = c; }   /* :273-278 */
        else cqs_oracle_heap_push_u64(h, id_rank ? (uint64_t)id_rank[c] : c, score[c]);
    }
    uint64_t* tmp = (uint64_t*)malloc((k ? k : 1) * sizeof(uint64_t));
    const size_t cnt = cqs_oracle_heap_into_sorted(h, tmp, scores_out, k); /* :281 */
    if (ids) {
        for (size_t i = 0; i < cnt; ++i) chunks_out[i] = pushed[tmp[i]];
    } else if (id_rank) {
        uint64_t* of_rank = (uint64_t*)malloc((size_t)s->n * sizeof(uint64_t));
        for (uint64_t c = 0; c < s->n; ++c) of_rank[id_rank[c]] = c;
        for (size_t i = 0; i < cnt; ++i) chunks_out[i] = of_rank[tmp[i]];
        free(of_rank);
    } else {
        for (size_t i = 0; i < cnt; ++i) chunks_out[i] = tmp[i];
    }
    free(tmp); free(pushed); free(score); free(scored);
    cqs_oracle_heap_free(h);
    return cnt;
}
