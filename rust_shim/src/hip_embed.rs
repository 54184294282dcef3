//! MI355X EmbeddingGemma-300m forward for cqs: the `session.run` block of `Embedder::embed_batch`
//! (src/embedder/core.rs:1091-1203) over libcqs_hip.so (C ABI: include/cqs_hip.h, embed section).
//!
//! Drop-in location: `src/embedder/hip.rs`, `#[cfg(feature = "hip-embed")] mod hip;` in
//! `src/embedder/mod.rs`, one `ExecutionProvider::Hip { device_id }` arm next to `ROCm`
//! (src/embedder/mod.rs:198-236) and, in `embed_batch`, the branch shown at the bottom of this file.
//! Everything around the forward stays as it is: tokenizer, prefixes, `embed_batch_size()` batching,
//! the LRU / disk caches, `pad_2d_i64_from_encodings`, and the per-row `normalize_l2`
//! (core.rs:1196-1203) - the library returns the pooled + Dense-projected vector NOT normalised, exactly
//! what the ONNX graph's `sentence_embedding` output holds.
//!
//! Not compiled here (no Rust toolchain in the build image); the same entry points are exercised by
//! `cqs_amd/embedder.py` and `tests/test_embed_gpu.py`.

use std::ffi::CString;
use std::os::raw::c_char;
use std::path::Path;

use super::EmbedderError;

// ---- C ABI (include/cqs_hip.h) ------------------------------------------------
#[repr(C)]
struct CqsHipEmbedder {
    _private: [u8; 0],
}

/// `cqs_hip_embed_config` - field order and types as in the header.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct CqsHipEmbedConfig {
    pub vocab_size: u32,
    pub hidden: u32,
    pub layers: u32,
    pub heads: u32,
    pub kv_heads: u32,
    pub head_dim: u32,
    pub intermediate: u32,
    pub dense_hidden: u32,
    pub sliding_window: u32,
    pub sliding_pattern: u32,
    pub max_seq: u32,
    pub rms_eps: f32,
    pub rope_theta_global: f32,
    pub rope_theta_local: f32,
    pub query_pre_attn_scalar: f32,
}

const CQS_HIP_OK: i32 = 0;

#[link(name = "cqs_hip")]
extern "C" {
    fn cqs_hip_embed_config_default(cfg: *mut CqsHipEmbedConfig);
    fn cqs_hip_embedder_load_dir(
        model_dir: *const c_char,
        cfg: *const CqsHipEmbedConfig,
        device: i32,
        out: *mut *mut CqsHipEmbedder,
    ) -> i32;
    fn cqs_hip_embedder_destroy(e: *mut CqsHipEmbedder);
    fn cqs_hip_embedder_dim(e: *const CqsHipEmbedder) -> u32;
    fn cqs_hip_embedder_max_seq(e: *const CqsHipEmbedder) -> u32;
    fn cqs_hip_embedder_poisoned(e: *const CqsHipEmbedder) -> i32;
    fn cqs_hip_embedder_last_error(e: *const CqsHipEmbedder, buf: *mut c_char, cap: usize) -> usize;
    // `Embedder::warm()` (src/embedder/core.rs:933-957): build + replay the search-time chain's graphs for every query
    // length up to `max_tokens` before the first real query; the counters say which path queries really take
    fn cqs_hip_embedder_warm(e: *mut CqsHipEmbedder, max_tokens: u32) -> i32;
    fn cqs_hip_embedder_query_graph_stats(
        e: *const CqsHipEmbedder,
        captured: *mut u64,
        failed: *mut u64,
        replays: *mut u64,
        eager: *mut u64,
    );
    fn cqs_hip_embed(
        e: *mut CqsHipEmbedder,
        input_ids: *const i64,
        attention_mask: *const i64,
        batch: u32,
        seq_len: u32,
        out: *mut f32,
    ) -> i32;
    // asynchronous form (up to 3 tickets in flight): the index pipeline keeps the device busy while the host
    // tokenises / pads the next `embed_batch_size()` chunk
    fn cqs_hip_embed_submit(
        e: *mut CqsHipEmbedder,
        input_ids: *const i64,
        attention_mask: *const i64,
        batch: u32,
        seq_len: u32,
        ticket: *mut u64,
    ) -> i32;
    fn cqs_hip_embed_collect(e: *mut CqsHipEmbedder, ticket: u64, out: *mut f32) -> i32;
}

/// The "session" of the Hip execution provider: owns the device weights of one model.
/// `Send` (moved into the embedder's `Mutex<Option<..>>` like the ort `Session`,
/// src/embedder/core.rs session()); calls are serialised by that mutex and again inside the library.
pub struct HipEmbedSession {
    handle: *mut CqsHipEmbedder,
    dim: usize,
}

unsafe impl Send for HipEmbedSession {}

impl HipEmbedSession {
    /// Replaces `create_session(model_path, provider)` (src/embedder/provider.rs:349-447).
    /// `model_dir` is what `ensure_model` resolves (src/embedder/download.rs:9-136): the directory that holds
    /// `onnx/model.onnx` + `onnx/model.onnx_data` (or the flat `model.onnx`) - the library parses the ONNX
    /// initialisers itself; a Hugging Face checkpoint directory (`model.safetensors`, `2_Dense/`, `3_Dense/`) works too.
    pub fn open(model_dir: &Path, device_id: i32) -> Result<Self, EmbedderError> {
        let dir = CString::new(model_dir.to_string_lossy().as_bytes())
            .map_err(|e| EmbedderError::InferenceFailed(format!("model dir: {e}")))?;
        let mut cfg = std::mem::MaybeUninit::<CqsHipEmbedConfig>::uninit();
        let mut handle: *mut CqsHipEmbedder = std::ptr::null_mut();
        // SAFETY: cfg is fully written by the callee; handle is an out pointer.
        let rc = unsafe {
            cqs_hip_embed_config_default(cfg.as_mut_ptr());
            cqs_hip_embedder_load_dir(dir.as_ptr(), cfg.as_ptr(), device_id, &mut handle)
        };
        if rc != CQS_HIP_OK || handle.is_null() {
            return Err(EmbedderError::InferenceFailed(format!(
                "cqs_hip_embedder_load_dir({}) failed: status {rc}",
                model_dir.display()
            )));
        }
        let dim = unsafe { cqs_hip_embedder_dim(handle) } as usize;
        Ok(Self { handle, dim })
    }

    pub fn embedding_dim(&self) -> usize {
        self.dim
    }

    pub fn max_seq(&self) -> usize {
        unsafe { cqs_hip_embedder_max_seq(self.handle) as usize }
    }

    /// A device failure poisons the handle; the embedder drops the session and re-opens it, the way
    /// `clear_session` (src/embedder/core.rs) handles a wedged ort session.
    pub fn is_poisoned(&self) -> bool {
        unsafe { cqs_hip_embedder_poisoned(self.handle) != 0 }
    }

    fn last_error(&self) -> String {
        let mut buf = vec![0u8; 512];
        let n = unsafe { cqs_hip_embedder_last_error(self.handle, buf.as_mut_ptr() as *mut c_char, buf.len()) };
        buf.truncate(n.min(buf.len()));
        String::from_utf8_lossy(&buf).into_owned()
    }

    /// What `Embedder::warm()` (core.rs:933-957) calls before its dummy inference: every `embed_query` length up to
    /// 128 tokens then replays a ready hipGraph (first call within a few percent of steady state) instead of paying
    /// an eager chain + capture + instantiate on the first query of each length.  Takes about half a second.
    pub fn warm(&mut self) -> Result<(), EmbedderError> {
        // SAFETY: the handle is live for the lifetime of self.
        let rc = unsafe { cqs_hip_embedder_warm(self.handle, 128) };
        if rc != CQS_HIP_OK {
            return Err(EmbedderError::InferenceFailed(format!("cqs_hip_embedder_warm failed: status {rc}: {}", self.last_error())));
        }
        let (mut captured, mut failed, mut replays, mut eager) = (0u64, 0u64, 0u64, 0u64);
        // SAFETY: four valid out-pointers.
        unsafe { cqs_hip_embedder_query_graph_stats(self.handle, &mut captured, &mut failed, &mut replays, &mut eager) };
        if failed > 0 {
            tracing::warn!(captured, failed, detail = %self.last_error(), "HIP embedder: some query lengths run without a captured graph");
        } else {
            tracing::debug!(captured, replays, eager, "HIP embedder warmed");
        }
        Ok(())
    }

    /// The `session.run` replacement: `input_ids` / `attention_mask` are the row-major `[batch, max_len]`
    /// i64 arrays `pad_2d_i64_from_encodings` builds (core.rs:1030-1035).  Returns `batch` rows of
    /// `embedding_dim()` floats, pooled and projected, NOT normalised.
    pub fn run(
        &mut self,
        input_ids: &[i64],
        attention_mask: &[i64],
        batch: usize,
        max_len: usize,
    ) -> Result<Vec<f32>, EmbedderError> {
        if input_ids.len() != batch * max_len || attention_mask.len() != batch * max_len {
            return Err(EmbedderError::InferenceFailed("hip embed: shape mismatch".into()));
        }
        let mut out = vec![0f32; batch * self.dim];
        // SAFETY: the slices outlive the call; the library copies what it needs before returning.
        let rc = unsafe {
            cqs_hip_embed(
                self.handle,
                input_ids.as_ptr(),
                attention_mask.as_ptr(),
                batch as u32,
                max_len as u32,
                out.as_mut_ptr(),
            )
        };
        if rc != CQS_HIP_OK {
            return Err(EmbedderError::InferenceFailed(format!(
                "cqs_hip_embed failed: status {rc}: {}",
                self.last_error()
            )));
        }
        Ok(out)
    }
}

impl HipEmbedSession {
    /// `embed_documents` (src/embedder/core.rs:718-751) hands `embed_batch` one `embed_batch_size()` chunk after the
    /// other and waits for each.  With the Hip provider the chunks can overlap: `submit` returns as soon as the batch
    /// is packed into pinned staging and enqueued, `collect` waits for one ticket.  Keep at most 3 in flight.
    pub fn submit(&mut self, input_ids: &[i64], attention_mask: &[i64], batch: usize, max_len: usize) -> Result<(u64, usize), EmbedderError> {
        if input_ids.len() != batch * max_len || attention_mask.len() != batch * max_len {
            return Err(EmbedderError::InferenceFailed("hip embed: shape mismatch".into()));
        }
        let mut ticket = 0u64;
        let rc = unsafe {
            cqs_hip_embed_submit(self.handle, input_ids.as_ptr(), attention_mask.as_ptr(), batch as u32, max_len as u32, &mut ticket)
        };
        if rc != CQS_HIP_OK {
            return Err(EmbedderError::InferenceFailed(format!("cqs_hip_embed_submit failed: status {rc}: {}", self.last_error())));
        }
        Ok((ticket, batch))
    }

    pub fn collect(&mut self, ticket: (u64, usize)) -> Result<Vec<f32>, EmbedderError> {
        let mut out = vec![0f32; ticket.1 * self.dim];
        let rc = unsafe { cqs_hip_embed_collect(self.handle, ticket.0, out.as_mut_ptr()) };
        if rc != CQS_HIP_OK {
            return Err(EmbedderError::InferenceFailed(format!("cqs_hip_embed_collect failed: status {rc}: {}", self.last_error())));
        }
        Ok(out)
    }

    /// Give a ticket's submission slot back without taking its rows (`out` = NULL): what a caller that bails out of
    /// an index run - an `Err` from a later `submit`, a cancelled pipeline - owes every ticket it still holds; a slot
    /// is released only by a collect, and a non-device failure does not poison the engine.
    pub fn abandon(&mut self, ticket: (u64, usize)) {
        // SAFETY: NULL `out` is the documented abandon form; the status is irrelevant (a failed collect frees the slot too).
        let _ = unsafe { cqs_hip_embed_collect(self.handle, ticket.0, std::ptr::null_mut()) };
    }
}

impl Drop for HipEmbedSession {
    fn drop(&mut self) {
        // SAFETY: handle came from cqs_hip_embedder_load_dir and is destroyed exactly once.
        unsafe { cqs_hip_embedder_destroy(self.handle) }
    }
}

// ---- the branch in Embedder::embed_batch (replacing core.rs:1040-1195 when the provider is Hip) ----
//
//     let input_ids_arr = pad_2d_i64_from_encodings(&encodings, |e| e.get_ids(), max_len, input_pad_id);
//     let attention_mask_arr = pad_2d_i64_from_encodings(&encodings, |e| e.get_attention_mask(), max_len, 0);
//     if let ExecutionProvider::Hip { .. } = self.provider {
//         let mut guard = self.hip_session()?;                       // Mutex<Option<HipEmbedSession>>, lazy open
//         let session = guard.as_mut().expect("hip_session() guarantees initialized");
//         let flat = session.run(
//             input_ids_arr.as_slice().expect("standard layout"),
//             attention_mask_arr.as_slice().expect("standard layout"),
//             texts.len(),
//             max_len,
//         )?;
//         return Ok(flat
//             .chunks_exact(session.embedding_dim())
//             .map(|row| Embedding::new(normalize_l2(row.to_vec())))   // core.rs:1196-1203
//             .collect());
//     }
