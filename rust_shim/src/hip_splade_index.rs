//! MI355X sparse retrieval leg over libcqs_hip.so (C ABI: include/cqs_hip.h, "sparse index"): the in-HBM replacement of
//! `SpladeIndex` (src/splade/index.rs:177-306) - same constructor input, same `search` / `search_with_filter` / `len` /
//! `is_empty` / `unique_tokens`, same result order (score desc, id asc) and bit-identical scores.
//!
//! Drop-in: `src/splade/hip_index.rs` behind the `hip-aux` feature.  `CommandContext::splade_index()`
//! (src/cli/store.rs:367-400) and `BatchView::ensure_splade_index()` (src/cli/batch/view.rs:842-908) build it from the same
//! `(chunk_id, SparseVector)` rows they hand to `SpladeIndex::load_or_build`; `search_hybrid_inner`
//! (src/search/query.rs:898-901) calls `search_with_filter` on it unchanged - the fusion (:909-1010) stays in Rust.
//!
//! Not compiled here (no Rust toolchain in the build image); the same entry points are exercised by
//! `cqs_amd/splade_index.py` and `tests/test_sparse_index_gpu.py`.

use std::os::raw::c_char;

use crate::index::IndexResult;
use crate::splade::SparseVector;

#[repr(C)]
struct CqsHipSparseIndex {
    _private: [u8; 0],
}

const CQS_HIP_OK: i32 = 0;
/// `CQS_HIP_MAX_K`: the largest k one call serves (production asks for candidate_count_for(limit) >= 500).
pub const MAX_K: usize = 1024;

#[link(name = "cqs_hip")]
extern "C" {
    fn cqs_hip_sparse_index_create(
        doc_off: *const u64,
        tokens: *const u32,
        weights: *const f32,
        n: u64,
        id_rank: *const u32,
        device: i32,
        out: *mut *mut CqsHipSparseIndex,
    ) -> i32;
    fn cqs_hip_sparse_index_create_inverted(
        token_ids: *const u32,
        list_off: *const u64,
        post_chunks: *const u32,
        post_weights: *const f32,
        n_tokens: u64,
        n: u64,
        id_rank: *const u32,
        device: i32,
        out: *mut *mut CqsHipSparseIndex,
    ) -> i32;
    fn cqs_hip_sparse_index_save(idx: *mut CqsHipSparseIndex, path: *const c_char, generation: u64, out_checksum: *mut u64) -> i32;
    fn cqs_hip_sparse_index_load(path: *const c_char, expected_chunks: u64, generation: u64, device: i32, out: *mut *mut CqsHipSparseIndex) -> i32;
    fn cqs_hip_sparse_index_destroy(idx: *mut CqsHipSparseIndex);
    fn cqs_hip_sparse_index_len(idx: *const CqsHipSparseIndex) -> u64;
    fn cqs_hip_sparse_index_unique_tokens(idx: *const CqsHipSparseIndex) -> u64;
    fn cqs_hip_sparse_index_postings(idx: *const CqsHipSparseIndex) -> u64;
    fn cqs_hip_sparse_index_search(
        idx: *mut CqsHipSparseIndex,
        q_tokens: *const u32,
        q_weights: *const f32,
        n_terms: u32,
        k: u32,
        keep_bitset: *const u32,
        out_chunks: *mut u64,
        out_scores: *mut f32,
        out_count: *mut u32,
    ) -> i32;
    fn cqs_hip_sparse_index_search_batch(
        idx: *mut CqsHipSparseIndex,
        q_off: *const u64,
        q_tokens: *const u32,
        q_weights: *const f32,
        b: u32,
        k: u32,
        keep_bitset: *const u32,
        out_chunks: *mut u64,
        out_scores: *mut f32,
        out_counts: *mut u32,
    ) -> i32;
    fn cqs_hip_sparse_index_last_search(idx: *const CqsHipSparseIndex, accumulate_ms: *mut f32, touched_postings: *mut u64) -> i32;
    fn cqs_hip_sparse_index_poisoned(idx: *const CqsHipSparseIndex) -> i32;
    fn cqs_hip_sparse_index_last_error(idx: *const CqsHipSparseIndex, buf: *mut c_char, cap: usize) -> usize;
}

/// The sparse retrieval leg with its postings resident on the GPU (`SpladeIndex`'s surface).
pub struct HipSpladeIndex {
    raw: *mut CqsHipSparseIndex,
    /// chunk id of every position of the build order - the library answers in positions.
    id_map: Vec<Box<str>>,
}

// The handle serialises device work behind its own mutex (include/cqs_hip.h, threading note).
unsafe impl Send for HipSpladeIndex {}
unsafe impl Sync for HipSpladeIndex {}

impl HipSpladeIndex {
    /// `SpladeIndex::build` (index.rs:191-212).  `None` = no usable device / build refused: the caller keeps the CPU index.
    pub fn build(chunks: Vec<(String, SparseVector)>, device: i32) -> Option<Self> {
        let _span = tracing::info_span!("hip_splade_index_build", chunks = chunks.len()).entered();
        let n = chunks.len();
        let mut doc_off: Vec<u64> = Vec::with_capacity(n + 1);
        doc_off.push(0);
        let total: usize = chunks.iter().map(|(_, v)| v.len()).sum();
        let mut tokens: Vec<u32> = Vec::with_capacity(total);
        let mut weights: Vec<f32> = Vec::with_capacity(total);
        let mut id_map: Vec<Box<str>> = Vec::with_capacity(n);
        for (chunk_id, sparse) in chunks {
            for &(token_id, weight) in &sparse {
                tokens.push(token_id);
                weights.push(weight);
            }
            doc_off.push(tokens.len() as u64);
            id_map.push(chunk_id.into_boxed_str());
        }
        // rank of every id in ascending order (ties in chunk order): the order BoundedScoreHeap breaks score ties in
        let mut order: Vec<u32> = (0..n as u32).collect();
        order.sort_by(|&a, &b| id_map[a as usize].cmp(&id_map[b as usize]).then(a.cmp(&b)));
        let mut id_rank = vec![0u32; n];
        for (r, &i) in order.iter().enumerate() {
            id_rank[i as usize] = r as u32;
        }
        let mut raw: *mut CqsHipSparseIndex = std::ptr::null_mut();
        let rc = unsafe {
            cqs_hip_sparse_index_create(doc_off.as_ptr(), tokens.as_ptr(), weights.as_ptr(), n as u64, id_rank.as_ptr(), device, &mut raw)
        };
        if rc != CQS_HIP_OK || raw.is_null() {
            tracing::warn!(rc, "HIP SPLADE index build failed, keeping the in-memory index");
            return None;
        }
        tracing::info!(chunks = n, postings = total, "HIP SPLADE index built");
        Some(Self { raw, id_map })
    }

    /// From an index the CPU side already holds - e.g. right after `SpladeIndex::load` read `splade.index.bin` - without
    /// going back to the store for the rows: the postings map and the id map as they are.
    pub fn from_postings(postings: &std::collections::HashMap<u32, Vec<(usize, f32)>>, id_map: &[Box<str>], device: i32) -> Option<Self> {
        let n = id_map.len();
        let mut token_ids: Vec<u32> = Vec::with_capacity(postings.len());
        let mut list_off: Vec<u64> = Vec::with_capacity(postings.len() + 1);
        list_off.push(0);
        let total: usize = postings.values().map(|l| l.len()).sum();
        let mut chunks: Vec<u32> = Vec::with_capacity(total);
        let mut weights: Vec<f32> = Vec::with_capacity(total);
        for (&token, list) in postings {
            token_ids.push(token);
            for &(chunk_idx, weight) in list {
                chunks.push(u32::try_from(chunk_idx).unwrap_or(u32::MAX)); // out of range: dropped by the library like index.rs:252
                weights.push(weight);
            }
            list_off.push(chunks.len() as u64);
        }
        let mut order: Vec<u32> = (0..n as u32).collect();
        order.sort_by(|&a, &b| id_map[a as usize].cmp(&id_map[b as usize]).then(a.cmp(&b)));
        let mut id_rank = vec![0u32; n];
        for (r, &i) in order.iter().enumerate() {
            id_rank[i as usize] = r as u32;
        }
        let mut raw: *mut CqsHipSparseIndex = std::ptr::null_mut();
        let rc = unsafe {
            cqs_hip_sparse_index_create_inverted(
                token_ids.as_ptr(), list_off.as_ptr(), chunks.as_ptr(), weights.as_ptr(), token_ids.len() as u64, n as u64,
                id_rank.as_ptr(), device, &mut raw,
            )
        };
        if rc != CQS_HIP_OK || raw.is_null() {
            tracing::warn!(rc, "HIP SPLADE index build from postings failed, keeping the in-memory index");
            return None;
        }
        Some(Self { raw, id_map: id_map.to_vec() })
    }

    /// `SpladeIndex::save` (index.rs:346): the device-ready arrays under `path` (own format), tied to `generation`.
    pub fn save(&self, path: &std::path::Path, generation: u64) -> bool {
        let Ok(c) = std::ffi::CString::new(path.to_string_lossy().as_bytes()) else { return false };
        let mut ck = 0u64;
        unsafe { cqs_hip_sparse_index_save(self.raw, c.as_ptr(), generation, &mut ck) == CQS_HIP_OK }
    }

    /// `SpladeIndex::load` (index.rs:677): `None` = missing / stale generation / damaged - rebuild.  The id map comes from
    /// the store (`chunk ids in index order`), as for the dense index's sidecar.
    pub fn load(path: &std::path::Path, generation: u64, id_map: Vec<Box<str>>, device: i32) -> Option<Self> {
        let c = std::ffi::CString::new(path.to_string_lossy().as_bytes()).ok()?;
        let mut raw: *mut CqsHipSparseIndex = std::ptr::null_mut();
        let rc = unsafe { cqs_hip_sparse_index_load(c.as_ptr(), id_map.len() as u64, generation, device, &mut raw) };
        if rc != CQS_HIP_OK || raw.is_null() {
            return None;
        }
        Some(Self { raw, id_map })
    }

    /// Largest `k` one call serves (the library's top-k capacity).  The reference asks the sparse leg for
    /// `candidate_count_for(limit) = max(500, 5 * limit)` with no cap (query.rs:858, 896-897; limits.rs:315-320), i.e. more
    /// than this for `limit > 204`: the caller caps to it the way `cap_k_to_backend` caps the dense leg (query.rs:232-245),
    /// or keeps the in-memory `SpladeIndex` for such limits.  `search*` below clamp and SAY SO.
    pub fn max_k(&self) -> usize {
        MAX_K
    }

    fn clamp_k(&self, k: usize) -> usize {
        if k > MAX_K {
            tracing::warn!(requested = k, served = MAX_K, "HIP SPLADE index: k above max_k, sparse pool is shorter than SpladeIndex::search would return");
        }
        k.min(MAX_K)
    }

    /// `SpladeIndex::search` (index.rs:214-216): every chunk is a candidate.  No bitset is built or uploaded (round 4 went
    /// through `search_with_filter(|_| true)`: one closure call per chunk and a 125 KB upload in front of a 50 us search).
    pub fn search(&self, query: &SparseVector, k: usize) -> Vec<IndexResult> {
        self.search_keep(query, k, None)
    }

    /// `SpladeIndex::search_with_filter` (index.rs:223-290).  The predicate is evaluated once per chunk id on the host
    /// and handed over as a bitset; a device failure logs and returns no results (the caller's dense leg still answers).
    /// A caller that searches repeatedly under one filter builds the bitset once (`keep_bitset`) and calls
    /// `search_with_keep`: the reference evaluates its predicate on touched postings only, this path cannot.
    pub fn search_with_filter(&self, query: &SparseVector, k: usize, filter: &dyn Fn(&str) -> bool) -> Vec<IndexResult> {
        if query.is_empty() || self.id_map.is_empty() || k == 0 {
            return Vec::new();
        }
        match self.keep_bitset(filter) {
            Some(keep) => self.search_keep(query, k, Some(&keep)),
            None => self.search_keep(query, k, None),          // the predicate kept everything
        }
    }

    /// The keep-bitset of a predicate over this index's chunk ids (bit i = chunk i kept); `None` when it keeps every
    /// chunk.  Cache it per filter key: it costs one predicate call per chunk.
    pub fn keep_bitset(&self, filter: &dyn Fn(&str) -> bool) -> Option<Vec<u32>> {
        let mut keep = vec![0u32; (self.id_map.len() + 31) / 32];
        let mut all = true;
        for (i, id) in self.id_map.iter().enumerate() {
            if filter(id) {
                keep[i / 32] |= 1 << (i % 32);
            } else {
                all = false;
            }
        }
        if all { None } else { Some(keep) }
    }

    /// `search_with_filter` with a bitset from `keep_bitset`.
    pub fn search_with_keep(&self, query: &SparseVector, k: usize, keep: &[u32]) -> Vec<IndexResult> {
        if keep.len() != (self.id_map.len() + 31) / 32 {
            tracing::warn!(words = keep.len(), "HIP SPLADE index: keep bitset of the wrong length, no results");
            return Vec::new();
        }
        self.search_keep(query, k, Some(keep))
    }

    fn search_keep(&self, query: &SparseVector, k: usize, keep: Option<&[u32]>) -> Vec<IndexResult> {
        let _span = tracing::debug_span!("hip_splade_index_search", k, query_terms = query.len(), index_size = self.id_map.len()).entered();
        if query.is_empty() || self.id_map.is_empty() || k == 0 {
            return Vec::new();
        }
        let k = self.clamp_k(k);
        let q_tokens: Vec<u32> = query.iter().map(|&(t, _)| t).collect();
        let q_weights: Vec<f32> = query.iter().map(|&(_, w)| w).collect();
        let mut chunks = vec![0u64; k];
        let mut scores = vec![0f32; k];
        let mut count = 0u32;
        let rc = unsafe {
            cqs_hip_sparse_index_search(
                self.raw,
                q_tokens.as_ptr(),
                q_weights.as_ptr(),
                q_tokens.len() as u32,
                k as u32,
                keep.map_or(std::ptr::null(), |w| w.as_ptr()),
                chunks.as_mut_ptr(),
                scores.as_mut_ptr(),
                &mut count,
            )
        };
        if rc != CQS_HIP_OK {
            tracing::warn!(rc, error = %self.last_error(), "HIP SPLADE search failed");
            return Vec::new();
        }
        (0..count as usize)
            .filter_map(|i| self.id_map.get(chunks[i] as usize).map(|id| IndexResult { id: id.to_string(), score: scores[i] }))
            .collect()
    }

    /// Several queries in one pair of launches (`cqs eval`, or a daemon that gathers its clients' queries): each answer is the
    /// one `search` gives for that query alone.  At most 64 queries per call; more are served in slices.
    pub fn search_batch(&self, queries: &[&SparseVector], k: usize) -> Vec<Vec<IndexResult>> {
        let k = self.clamp_k(k);
        let mut all = Vec::with_capacity(queries.len());
        for slice in queries.chunks(64) {
            let mut q_off: Vec<u64> = Vec::with_capacity(slice.len() + 1);
            q_off.push(0);
            let mut toks: Vec<u32> = Vec::new();
            let mut wts: Vec<f32> = Vec::new();
            for q in slice {
                toks.extend(q.iter().map(|&(t, _)| t));
                wts.extend(q.iter().map(|&(_, w)| w));
                q_off.push(toks.len() as u64);
            }
            let b = slice.len();
            let mut chunks = vec![0u64; b * k.max(1)];
            let mut scores = vec![0f32; b * k.max(1)];
            let mut counts = vec![0u32; b];
            let rc = unsafe {
                cqs_hip_sparse_index_search_batch(
                    self.raw, q_off.as_ptr(), toks.as_ptr(), wts.as_ptr(), b as u32, k as u32, std::ptr::null(),
                    chunks.as_mut_ptr(), scores.as_mut_ptr(), counts.as_mut_ptr(),
                )
            };
            if rc != CQS_HIP_OK {
                tracing::warn!(rc, error = %self.last_error(), "HIP SPLADE batch search failed");
                all.extend((0..b).map(|_| Vec::new()));
                continue;
            }
            for q in 0..b {
                all.push(
                    (0..counts[q] as usize)
                        .filter_map(|i| self.id_map.get(chunks[q * k + i] as usize).map(|id| IndexResult { id: id.to_string(), score: scores[q * k + i] }))
                        .collect(),
                );
            }
        }
        all
    }

    /// `SpladeIndex::len` (index.rs:294-296).
    pub fn len(&self) -> usize {
        unsafe { cqs_hip_sparse_index_len(self.raw) as usize }
    }

    /// `SpladeIndex::is_empty` (index.rs:299-301).
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }

    /// `SpladeIndex::unique_tokens` (index.rs:304-306): tokens that own a posting list.
    pub fn unique_tokens(&self) -> usize {
        unsafe { cqs_hip_sparse_index_unique_tokens(self.raw) as usize }
    }

    pub fn postings(&self) -> u64 {
        unsafe { cqs_hip_sparse_index_postings(self.raw) }
    }

    pub fn is_poisoned(&self) -> bool {
        unsafe { cqs_hip_sparse_index_poisoned(self.raw) != 0 }
    }

    /// (device milliseconds of the last search's scoring launches, postings they read)
    pub fn last_search(&self) -> (f32, u64) {
        let (mut ms, mut touched) = (0f32, 0u64);
        unsafe { cqs_hip_sparse_index_last_search(self.raw, &mut ms, &mut touched) };
        (ms, touched)
    }

    fn last_error(&self) -> String {
        let mut buf = vec![0u8; 512];
        let n = unsafe { cqs_hip_sparse_index_last_error(self.raw, buf.as_mut_ptr() as *mut c_char, buf.len()) };
        buf.truncate(n.min(511));
        String::from_utf8_lossy(&buf).into_owned()
    }
}

impl Drop for HipSpladeIndex {
    fn drop(&mut self) {
        unsafe { cqs_hip_sparse_index_destroy(self.raw) }
    }
}
