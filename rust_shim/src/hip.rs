//! MI355X exact GPU index for cqs: `impl VectorIndex` / `impl IndexBackend` over
//! libcqs_hip.so (C ABI: include/cqs_hip.h).
//!
//! Drop-in location: `src/hip.rs`, `#[cfg(feature = "hip-index")] pub mod hip;` in
//! `src/lib.rs`, one row in the backend table (`src/index.rs:338-345`):
//!
//! ```ignore
//! crate::hip::HipBackend, cfg(feature = "hip-index");
//! ```
//!
//! Multi-GPU: `CQS_HIP_DEVICES=0,1,..` makes the ONE daemon process shard the corpus over those devices through
//! `cqs_hip_index_create_sharded` (per-device scan, RCCL all-gather of candidates, host merge inside the library); the
//! trait surface does not change.  Persistence: `index.hipflat` + `index.hipflat.meta`, persisted-first open.
//!
//! Shape follows `CagraIndex` (src/cagra.rs:255-277): the index owns `id_map`
//! (row -> chunk id, rowid order); the device side is addressed by row number.
//! Conventions kept from the reference (src/cagra.rs:445-470, 543-626, 1715-1800):
//! `search` never panics and never returns an error for device trouble - it logs and
//! returns an empty Vec; `try_open` returns `Ok(None)` to fall through to the next
//! backend and `Err` only for store errors.

use std::os::raw::c_char;
use std::sync::atomic::{AtomicBool, Ordering};

use crate::embedder::Embedding;
use crate::index::{BackendContext, DistanceMetric, IndexBackend, IndexResult, VectorIndex};
use crate::store::{ClearHnswDirty, Store, StoreError};

// ---- C ABI (include/cqs_hip.h) ------------------------------------------------
#[repr(C)]
struct CqsHipIndex {
    _private: [u8; 0],
}

const CQS_HIP_OK: i32 = 0;
const CQS_HIP_METRIC_COSINE: u32 = 0;
const CQS_HIP_METRIC_DOT: u32 = 1;
const CQS_HIP_MODE_RAW: u32 = 0;

#[link(name = "cqs_hip")]
extern "C" {
    fn cqs_hip_device_count() -> i32;
    fn cqs_hip_device_mem(device: i32, free_bytes: *mut u64, total_bytes: *mut u64) -> i32;
    fn cqs_hip_index_create(
        rows: *const f32,
        n: u64,
        dim: u32,
        metric: u32,
        device: i32,
        row_base: u64,
        out: *mut *mut CqsHipIndex,
    ) -> i32;
    // ONE process, several GPUs (north_star): rows cut over `devices`, RCCL all-gather of per-shard candidates inside
    // the library, host merge; every other entry point takes the returned handle unchanged
    fn cqs_hip_index_create_sharded(
        rows: *const f32,
        n: u64,
        dim: u32,
        metric: u32,
        devices: *const i32,
        n_devices: u32,
        row_base: u64,
        out: *mut *mut CqsHipIndex,
    ) -> i32;
    fn cqs_hip_index_load_sharded(path: *const c_char, expected_dim: u32, expected_rows: u64, devices: *const i32,
                                  n_devices: u32, row_base: u64, out: *mut *mut CqsHipIndex) -> i32;
    fn cqs_hip_index_neighbors(idx: *mut CqsHipIndex, target_row: u64, limit: u32, out_rows: *mut u64,
                               out_scores: *mut f32, out_count: *mut u32) -> i32;
    fn cqs_hip_index_metric(idx: *const CqsHipIndex) -> u32;
    fn cqs_hip_index_extend(idx: *mut CqsHipIndex, rows: *const f32, n_new: u64) -> i32;
    // persistence (the index.cagra + .meta analogue, src/cagra.rs:973-1157, 1174-1330): the blob is written /
    // validated by the library, the CagraMeta-style sidecar (magic, version, dim, chunk_count, id_map,
    // checksum, metric) by `HipIndex::save` / `HipIndex::load` below via serde_json like src/cagra.rs:1128-1147
    fn cqs_hip_index_save(idx: *mut CqsHipIndex, path: *const c_char, out_checksum: *mut u64) -> i32;
    fn cqs_hip_index_load(path: *const c_char, expected_dim: u32, expected_rows: u64, device: i32, row_base: u64,
                          out: *mut *mut CqsHipIndex) -> i32;
    fn cqs_hip_index_destroy(idx: *mut CqsHipIndex);
    fn cqs_hip_index_len(idx: *const CqsHipIndex) -> u64;
    fn cqs_hip_index_max_k(idx: *const CqsHipIndex) -> u32;
    fn cqs_hip_index_poisoned(idx: *const CqsHipIndex) -> i32;
    fn cqs_hip_index_last_error(idx: *const CqsHipIndex, buf: *mut c_char, cap: usize) -> usize;
    fn cqs_hip_index_search(
        idx: *mut CqsHipIndex,
        queries: *const f32,
        b: u32,
        query_dim: u32,
        k: u32,
        keep_bitset: *const u32,
        mode: u32,
        threshold: f32,
        out_rows: *mut u64,
        out_scores: *mut f32,
        out_counts: *mut u32,
    ) -> i32;
}

/// Exact brute-force GPU index (HBM-resident `[n, dim]` f32 + top-k on device).
pub struct HipIndex {
    handle: *mut CqsHipIndex,
    /// row -> chunk id, rowid order (same role as `CagraIndex::id_map`, src/cagra.rs:266).
    id_map: Vec<Box<str>>,
    dim: usize,
    metric: DistanceMetric,
    /// Set when the C side reports a device failure; `is_poisoned()` makes the daemon
    /// rebuild the index (src/index.rs:203-205, src/cagra.rs:472-489).
    poisoned: AtomicBool,
}

// SAFETY: the handle serialises device access behind its own mutex (include/cqs_hip.h,
// "Threading"); `id_map` is immutable after construction.  Same argument as
// `unsafe impl Send/Sync for CagraIndex` (src/cagra.rs:823-830).
unsafe impl Send for HipIndex {}
unsafe impl Sync for HipIndex {}

impl Drop for HipIndex {
    fn drop(&mut self) {
        // destroy() synchronises the index's streams first (src/cagra.rs:289-302)
        unsafe { cqs_hip_index_destroy(self.handle) }
    }
}

/// Sidecar of the persisted blob: the `CagraMeta` of this backend (src/cagra.rs:973-997) - the blob itself holds
/// only rows, so the row -> chunk-id map, the store's chunk count at save time and the blob's checksum live here.
#[derive(serde::Serialize, serde::Deserialize)]
struct HipMeta {
    magic: String,
    version: u32,
    dim: usize,
    chunk_count: usize,
    id_map: Vec<String>,
    checksum: String,
    metric: String,
}

const HIP_META_MAGIC: &str = "cqs-hip-flat-meta";
const HIP_META_VERSION: u32 = 1;

/// `CQS_HIP_DEVICES=0,1,2,3`: shard the corpus over these GPUs inside this process; unset / empty = device 0.
pub fn devices_from_env() -> Vec<i32> {
    let parsed: Vec<i32> = std::env::var("CQS_HIP_DEVICES")
        .ok()
        .map(|v| v.split(',').filter_map(|t| t.trim().parse().ok()).collect())
        .unwrap_or_default();
    if parsed.is_empty() { vec![0] } else { parsed }
}

fn hip_persist_enabled() -> bool {
    // same switch shape as CQS_CAGRA_PERSIST (src/cagra.rs:1013-1021): on unless set to 0 / false / off
    !matches!(std::env::var("CQS_HIP_PERSIST").as_deref(), Ok("0") | Ok("false") | Ok("off"))
}

impl HipIndex {
    /// `CagraIndex::gpu_available_for` analogue (src/cagra.rs:336-376): every listed device must exist and hold its
    /// shard (corpus / devices + score scratch headroom).
    pub fn gpu_available_for(n: usize, dim: usize, devices: &[i32]) -> bool {
        let count = unsafe { cqs_hip_device_count() };
        if count <= 0 || devices.is_empty() {
            return false;
        }
        // corpus + score/scratch headroom must fit HBM (288 GB on MI355X: no 2 GiB host cap,
        // unlike CQS_CAGRA_MAX_BYTES, src/cagra.rs:159-167)
        let per_shard = (n as u64).saturating_mul(dim as u64).saturating_mul(4).saturating_mul(5) / 4 / devices.len() as u64;
        devices.iter().all(|&d| {
            let (mut free, mut total) = (0u64, 0u64);
            d >= 0 && d < count && unsafe { cqs_hip_device_mem(d, &mut free, &mut total) } == CQS_HIP_OK && per_shard <= free
        })
    }

    /// `CagraIndex::build_from_flat` (src/cagra.rs:922-960).
    pub fn build_from_flat(
        id_map: Vec<String>,
        flat_data: Vec<f32>,
        dim: usize,
        metric: DistanceMetric,
        devices: &[i32],
    ) -> Result<Self, String> {
        if id_map.is_empty() || flat_data.len() != id_map.len() * dim || devices.is_empty() {
            return Err("HIP build: empty or misshapen dataset".into());
        }
        let mut handle: *mut CqsHipIndex = std::ptr::null_mut();
        let m = match metric {
            DistanceMetric::Cosine => CQS_HIP_METRIC_COSINE,
            DistanceMetric::DotProduct => CQS_HIP_METRIC_DOT,
        };
        let rc = unsafe {
            if devices.len() == 1 {
                cqs_hip_index_create(flat_data.as_ptr(), id_map.len() as u64, dim as u32, m, devices[0], 0, &mut handle)
            } else {
                cqs_hip_index_create_sharded(flat_data.as_ptr(), id_map.len() as u64, dim as u32, m, devices.as_ptr(),
                                             devices.len() as u32, 0, &mut handle)
            }
        };
        if rc != CQS_HIP_OK || handle.is_null() {
            return Err(format!("cqs_hip_index_create failed: {rc}"));
        }
        Ok(Self {
            handle,
            id_map: id_map.into_iter().map(String::into_boxed_str).collect(),
            dim,
            metric,
            poisoned: AtomicBool::new(false),
        })
    }

    /// Stream the store into a flat buffer (`build_from_store_with_metric`,
    /// src/cagra.rs:842-916).  Zero / non-finite rows are skipped exactly like
    /// `prepare_index_data` (src/hnsw/mod.rs:717-731) so `len()` and row->id agree
    /// with the HNSW backend.
    pub fn build_from_store<Mode>(store: &Store<Mode>, dim: usize, metric: DistanceMetric, devices: &[i32]) -> Result<Self, String> {
        let mut id_map: Vec<String> = Vec::new();
        let mut flat: Vec<f32> = Vec::new();
        let batch = crate::limits::dim_scaled_batch(10_000, dim, 500, 50_000);
        for batch_result in store.embedding_batches(batch) {
            let batch = batch_result.map_err(|e| format!("Failed to fetch batch: {e}"))?;
            for (chunk_id, embedding) in batch {
                let v = embedding.as_slice();
                if v.len() != dim {
                    return Err(format!("dimension mismatch: expected {dim}, got {}", v.len()));
                }
                if !v.iter().any(|x| *x != 0.0) || v.iter().any(|x| !x.is_finite()) {
                    tracing::warn!(chunk_id = %chunk_id, "Skipping zero / non-finite embedding");
                    continue;
                }
                id_map.push(chunk_id);
                flat.extend_from_slice(v);
            }
        }
        Self::build_from_flat(id_map, flat, dim, metric, devices)
    }

    /// Incremental add (the contract `tiered.rs` exposes, src/tiered.rs:1-43).
    pub fn extend(&mut self, ids: Vec<String>, rows: &[f32]) -> Result<(), String> {
        if rows.len() != ids.len() * self.dim {
            return Err("HIP extend: misshapen rows".into());
        }
        let rc = unsafe { cqs_hip_index_extend(self.handle, rows.as_ptr(), ids.len() as u64) };
        if rc != CQS_HIP_OK {
            return Err(self.last_error());
        }
        self.id_map.extend(ids.into_iter().map(String::into_boxed_str));
        Ok(())
    }

    /// `CagraIndex::save` (src/cagra.rs:1086-1157): the library streams the rows HBM -> `<path>.tmp` -> fsync -> `.bak`
    /// swap -> rename (save_blob_atomic_with_rollback, src/cagra.rs:1468-1592, restated in C++) and returns the blob's
    /// checksum; the sidecar goes out through write-temp + rename like `write_meta_atomic` (src/cagra.rs:1594-1652).
    /// A blob without its sidecar is useless, so a sidecar failure removes both (src/cagra.rs:1143-1147).
    pub fn save(&self, path: &std::path::Path, chunk_count: usize) -> Result<(), String> {
        let cpath = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|e| e.to_string())?;
        let mut checksum = 0u64;
        let rc = unsafe { cqs_hip_index_save(self.handle, cpath.as_ptr(), &mut checksum) };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_index_save failed ({rc}): {}", self.last_error()));
        }
        let meta = HipMeta {
            magic: HIP_META_MAGIC.into(),
            version: HIP_META_VERSION,
            dim: self.dim,
            chunk_count,
            id_map: self.id_map.iter().map(|s| s.to_string()).collect(),
            checksum: format!("{checksum:016x}"),
            metric: self.metric.as_str().to_string(),
        };
        let meta_path = path.with_extension("hipflat.meta");
        let tmp = path.with_extension("hipflat.meta.tmp");
        let write = || -> std::io::Result<()> {
            let f = std::fs::File::create(&tmp)?;
            serde_json::to_writer(std::io::BufWriter::new(&f), &meta).map_err(std::io::Error::other)?;
            f.sync_all()?;
            std::fs::rename(&tmp, &meta_path)
        };
        write().map_err(|e| {
            let _ = std::fs::remove_file(&tmp);
            Self::delete_persisted(path);
            format!("HIP sidecar write failed: {e}")
        })
    }

    /// `CagraIndex::load` (src/cagra.rs:1174-1330): sidecar magic / version / dim / chunk_count must match the store,
    /// then the library validates the blob (header, size, checksum while streaming it to HBM) and the checksum must
    /// be the sidecar's.  Any mismatch is an `Err`: the caller deletes both files and rebuilds.
    pub fn load(path: &std::path::Path, dim: usize, chunk_count: usize, devices: &[i32]) -> Result<Self, String> {
        let meta_path = path.with_extension("hipflat.meta");
        let meta: HipMeta = serde_json::from_reader(std::io::BufReader::new(
            std::fs::File::open(&meta_path).map_err(|e| format!("HIP sidecar unreadable: {e}"))?,
        ))
        .map_err(|e| format!("HIP sidecar unparsable: {e}"))?;
        if meta.magic != HIP_META_MAGIC || meta.version != HIP_META_VERSION {
            return Err("HIP sidecar: bad magic / version".into());
        }
        if meta.dim != dim || meta.chunk_count != chunk_count || meta.id_map.len() > chunk_count {
            return Err("HIP sidecar: stale (dim / chunk_count mismatch)".into());
        }
        let metric: DistanceMetric = meta.metric.parse().map_err(|_| "HIP sidecar: bad metric".to_string())?;
        let cpath = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|e| e.to_string())?;
        let mut handle: *mut CqsHipIndex = std::ptr::null_mut();
        let rows = meta.id_map.len() as u64;   // zero / non-finite rows were skipped at build: rows <= chunk_count
        let rc = unsafe {
            if devices.len() <= 1 {
                cqs_hip_index_load(cpath.as_ptr(), dim as u32, rows, devices.first().copied().unwrap_or(0), 0, &mut handle)
            } else {
                cqs_hip_index_load_sharded(cpath.as_ptr(), dim as u32, rows, devices.as_ptr(), devices.len() as u32, 0, &mut handle)
            }
        };
        if rc != CQS_HIP_OK || handle.is_null() {
            return Err(format!("HIP blob rejected (rc = {rc})"));
        }
        let idx = Self {
            handle,
            id_map: meta.id_map.into_iter().map(String::into_boxed_str).collect(),
            dim,
            metric,
            poisoned: AtomicBool::new(false),
        };
        // the sidecar must describe THIS blob: its checksum is the one the library just verified against the rows
        let blob_metric = unsafe { cqs_hip_index_metric(idx.handle) };
        let want = match metric { DistanceMetric::Cosine => CQS_HIP_METRIC_COSINE, DistanceMetric::DotProduct => CQS_HIP_METRIC_DOT };
        if blob_metric != want || !blob_checksum_matches(path, &meta.checksum) {
            return Err("HIP sidecar does not match the blob".into());
        }
        Ok(idx)
    }

    /// `CagraIndex::delete_persisted` (src/cagra.rs:1332-1345).
    pub fn delete_persisted(path: &std::path::Path) {
        let _ = std::fs::remove_file(path);
        let _ = std::fs::remove_file(path.with_extension("hipflat.meta"));
    }

    /// `find_neighbors` (src/cli/commands/search/neighbors.rs:86-132) on the resident corpus: exact kNN of an indexed
    /// chunk, itself excluded, (score desc, id asc), limit clamped to [1, SIMILAR_LIMIT_MAX] inside the library.
    /// `None` = the chunk is not in this index (the caller falls back to the store scan).
    pub fn find_neighbors(&self, target_id: &str, limit: usize) -> Option<Vec<IndexResult>> {
        let row = self.id_map.iter().position(|id| &**id == target_id)? as u64;
        let (mut rows, mut scores, mut count) = (vec![0u64; 100], vec![0f32; 100], 0u32);
        let rc = unsafe {
            cqs_hip_index_neighbors(self.handle, row, limit.min(u32::MAX as usize) as u32, rows.as_mut_ptr(),
                                    scores.as_mut_ptr(), &mut count)
        };
        if rc != CQS_HIP_OK {
            tracing::warn!(error = %self.last_error(), rc, "HIP neighbors failed");
            return None;
        }
        Some((0..count as usize)
            .filter_map(|i| self.id_map.get(rows[i] as usize).map(|id| IndexResult { id: id.to_string(), score: scores[i] }))
            .collect())
    }

    fn last_error(&self) -> String {
        let mut buf = vec![0u8; 512];
        let n = unsafe { cqs_hip_index_last_error(self.handle, buf.as_mut_ptr() as *mut c_char, buf.len()) };
        String::from_utf8_lossy(&buf[..n]).into_owned()
    }

    fn search_impl(&self, query: &Embedding, k: usize, bitset: Option<&[u32]>) -> Vec<IndexResult> {
        let k = k.min(unsafe { cqs_hip_index_max_k(self.handle) } as usize);
        let mut rows = vec![0u64; k];
        let mut scores = vec![0f32; k];
        let mut count = 0u32;
        let rc = unsafe {
            cqs_hip_index_search(
                self.handle,
                query.as_slice().as_ptr(),
                1,
                query.len() as u32,
                k as u32,
                bitset.map_or(std::ptr::null(), |b| b.as_ptr()),
                CQS_HIP_MODE_RAW,
                0.0,
                rows.as_mut_ptr(),
                scores.as_mut_ptr(),
                &mut count,
            )
        };
        if rc != CQS_HIP_OK {
            if unsafe { cqs_hip_index_poisoned(self.handle) } != 0 {
                self.poisoned.store(true, Ordering::Release);
            }
            tracing::error!(error = %self.last_error(), rc, "HIP search failed");
            return Vec::new();
        }
        (0..count as usize)
            .filter_map(|i| {
                self.id_map.get(rows[i] as usize).map(|id| IndexResult {
                    id: id.to_string(),
                    // exact dot of unit vectors; cap the f32 overshoot like src/cagra.rs:656-661
                    score: match self.metric {
                        DistanceMetric::Cosine => scores[i].min(1.0),
                        DistanceMetric::DotProduct => scores[i],
                    },
                })
            })
            .collect()
    }
}

impl VectorIndex for HipIndex {
    /// Called concurrently from the daemon's client threads (src/cli/watch/daemon.rs:273) on one `Arc<dyn VectorIndex>`:
    /// the library combines single-query, unfiltered calls that meet on the handle - single-device or sharded - into shared
    /// passes over the corpus (include/cqs_hip.h, "Concurrent callers"), each caller still getting the bytes its lone call
    /// would.  `CQS_HIP_COMBINE_BITS=relaxed` in the daemon's environment (read when the index is opened) trades that
    /// bit-reproducibility for throughput past 8 callers (blocks of >= 9 on the matrix cores; scores within 2e-6).
    fn search(&self, query: &Embedding, k: usize) -> Vec<IndexResult> {
        let _span = tracing::debug_span!("hip_search", k).entered();
        if self.id_map.is_empty() || k == 0 {
            return Vec::new();
        }
        if query.len() != self.dim {
            tracing::warn!(expected_dim = self.dim, actual_dim = query.len(), "Query dimension mismatch");
            return Vec::new();
        }
        if self.poisoned.load(Ordering::Acquire) {
            return Vec::new();
        }
        // non-finite queries are rejected inside the C ABI as well (count 0)
        self.search_impl(query, k, None)
    }

    fn len(&self) -> usize {
        unsafe { cqs_hip_index_len(self.handle) as usize }
    }

    fn name(&self) -> &'static str {
        "HIP"
    }

    fn dim(&self) -> usize {
        self.dim
    }

    /// GPU-native filtered search: bitset built on the host from the predicate
    /// (src/cagra.rs:747-757); all-pass / none / k-cap handled by the C ABI
    /// (src/cagra.rs:760-775).
    fn search_with_filter(&self, query: &Embedding, k: usize, filter: &dyn Fn(&str) -> bool) -> Vec<IndexResult> {
        if self.id_map.is_empty() || k == 0 || query.len() != self.dim {
            return Vec::new();
        }
        let n = self.id_map.len();
        let mut bitset = vec![0u32; n.div_ceil(32)];
        for (i, id) in self.id_map.iter().enumerate() {
            if filter(id) {
                bitset[i / 32] |= 1u32 << (i % 32);
            }
        }
        self.search_impl(query, k, Some(&bitset))
    }

    fn is_poisoned(&self) -> bool {
        self.poisoned.load(Ordering::Acquire) || unsafe { cqs_hip_index_poisoned(self.handle) } != 0
    }

    fn max_k(&self) -> Option<usize> {
        Some(unsafe { cqs_hip_index_max_k(self.handle) } as usize)
    }

    /// Scores are the exact dot product of unit vectors = the cosine the brute-force path
    /// recomputes, so the BLOB refetch can be skipped (src/search/query.rs:1152-1172).
    fn index_scores_are_cosine(&self) -> bool {
        matches!(self.metric, DistanceMetric::Cosine)
    }
}

/// Blob header (include/cqs_hip.h, `cqs_hip_index_save`): magic[8] version dim metric pad (u32 x 4) rows checksum (u64 x 2).
fn blob_checksum_matches(path: &std::path::Path, want_hex: &str) -> bool {
    use std::io::Read;
    let mut head = [0u8; 40];
    match std::fs::File::open(path).and_then(|mut f| f.read_exact(&mut head)) {
        Ok(()) => &head[..8] == b"CQSHIPF1" && format!("{:016x}", u64::from_le_bytes(head[32..40].try_into().unwrap())) == want_hex,
        Err(_) => false,
    }
}

/// `IndexBackend` registration; priority above CAGRA (100) and tiered (150).
pub struct HipBackend;

impl<Mode: ClearHnswDirty> IndexBackend<Mode> for HipBackend {
    fn name(&self) -> &'static str {
        "hip"
    }

    fn priority(&self) -> i32 {
        200
    }

    fn try_open(&self, ctx: &BackendContext<'_, Mode>) -> Result<Option<Box<dyn VectorIndex>>, StoreError> {
        const HIP_THRESHOLD_DEFAULT: u64 = 5000; // same gate as CQS_CAGRA_THRESHOLD (src/cagra.rs:1683-1690)
        let threshold: u64 = std::env::var("CQS_HIP_THRESHOLD")
            .ok()
            .and_then(|v| v.parse().ok())
            .or_else(|| ctx.policy.and_then(|p| p.cagra_threshold))
            .unwrap_or(HIP_THRESHOLD_DEFAULT);
        let chunk_count = ctx.store.chunk_count().unwrap_or(0);
        let dim = ctx.store.dim();
        let devices = devices_from_env();
        let gpu_available = HipIndex::gpu_available_for(chunk_count as usize, dim, &devices);
        if chunk_count < threshold || !gpu_available {
            tracing::info!(backend = "hnsw", source = "hip-ineligible", chunk_count, threshold, dim, gpu_available,
                "Vector index backend selected");
            return Ok(None);
        }
        // persisted first (src/cagra.rs:1726-1752): streaming the blob into HBM beats re-reading every BLOB out of
        // SQLite; a stale / corrupt pair is deleted and rebuilt
        let blob = ctx.cqs_dir.join("index.hipflat");
        if hip_persist_enabled() && blob.exists() {
            match HipIndex::load(&blob, dim, chunk_count as usize, &devices) {
                Ok(idx) => {
                    tracing::info!(backend = "hip", source = "persisted", vectors = idx.len(), chunk_count, threshold,
                        "Vector index backend selected");
                    return Ok(Some(Box::new(idx) as Box<dyn VectorIndex>));
                }
                Err(e) => {
                    tracing::warn!(error = %e, path = %blob.display(), "HIP persisted load failed, rebuilding from store");
                    HipIndex::delete_persisted(&blob);
                }
            }
        }
        let metric = match DistanceMetric::from_env() {
            Ok(Some(m)) => m,
            Ok(None) => crate::hnsw::HnswIndex::stored_metric(ctx.cqs_dir, "index").unwrap_or(DistanceMetric::Cosine),
            Err(e) => {
                tracing::warn!(error = %e, "Invalid CQS_DISTANCE_METRIC - falling through");
                return Ok(None);
            }
        };
        match HipIndex::build_from_store(ctx.store, dim, metric, &devices) {
            Ok(idx) => {
                tracing::info!(backend = "hip", source = "rebuilt", vectors = idx.len(), chunk_count, devices = ?devices,
                    "Vector index backend selected");
                if hip_persist_enabled() {
                    if let Err(e) = idx.save(&blob, chunk_count as usize) {
                        tracing::warn!(error = %e, path = %blob.display(), "Failed to persist HIP index (will rebuild next restart)");
                    }
                }
                Ok(Some(Box::new(idx) as Box<dyn VectorIndex>))
            }
            Err(e) => {
                tracing::warn!(error = %e, "Failed to build HIP index, falling through");
                Ok(None)
            }
        }
    }
}
