//! MI355X forwards of cqs's two BERT-family auxiliary models over libcqs_hip.so (C ABI: include/cqs_hip.h, "BERT-family
//! auxiliary models"): the `session.run` of `SpladeEncoder::{encode, encode_batch}` (src/splade/mod.rs:595-760,
//! :774-1075) and of `Reranker::compute_scores_opt` (src/reranker.rs:343-533).
//!
//! Drop-in: `src/splade/hip.rs` / `src/reranker_hip.rs` behind a `hip-aux` feature; in `SpladeEncoder::new` and
//! `Reranker::session()` the `create_session(...)` call gets a sibling arm that opens `HipBert` on the same model
//! directory, and the two `session.run` blocks call `splade_dense` / `rerank_logits` below.  Everything around the
//! forward stays: tokenizer, truncation to `max_seq_len`, the vocab probe (`cqs_hip_bert_vocab`), the threshold
//! filter, the sigmoid, batching (`reranker_batch_size`), caches.
//!
//! Not compiled here (no Rust toolchain in the build image); the same entry points are exercised by
//! `cqs_amd/splade.py` and `tests/test_bert_gpu.py`.

use std::ffi::CString;
use std::os::raw::c_char;
use std::path::Path;

#[repr(C)]
struct CqsHipBert {
    _private: [u8; 0],
}

/// `cqs_hip_bert_config` - field order and types as in the header.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct CqsHipBertConfig {
    pub vocab_size: u32,
    pub hidden: u32,
    pub layers: u32,
    pub heads: u32,
    pub intermediate: u32,
    pub max_pos: u32,
    pub type_vocab: u32,
    pub num_labels: u32,
    pub head: u32,
    pub ln_eps: f32,
}

pub const HEAD_MLM: u32 = 0;
pub const HEAD_CLASSIFIER: u32 = 1;
/// Encoder only: the BERT-family embedder presets (e5-base, v9-200k, bge-large, bge-large-ft).
pub const HEAD_NONE: u32 = 2;
const CQS_HIP_OK: i32 = 0;

#[link(name = "cqs_hip")]
extern "C" {
    fn cqs_hip_bert_config_default(head: u32, out: *mut CqsHipBertConfig) -> i32;
    fn cqs_hip_bert_load_dir(dir: *const c_char, cfg: *const CqsHipBertConfig, device: i32, out: *mut *mut CqsHipBert) -> i32;
    fn cqs_hip_bert_destroy(e: *mut CqsHipBert);
    fn cqs_hip_splade_encode(e: *mut CqsHipBert, tokens: *const i32, lens: *const u32, batch: u32, out_dense: *mut f32) -> i32;
    fn cqs_hip_splade_encode_sparse(e: *mut CqsHipBert, tokens: *const i32, lens: *const u32, batch: u32, threshold: f32,
                                    cap: u32, out_ids: *mut u32, out_weights: *mut f32, out_counts: *mut u32) -> i32;
    fn cqs_hip_splade_submit_sparse(e: *mut CqsHipBert, tokens: *const i32, lens: *const u32, batch: u32, threshold: f32,
                                    cap: u32, ticket: *mut u64) -> i32;
    fn cqs_hip_splade_collect_sparse(e: *mut CqsHipBert, ticket: u64, out_ids: *mut u32, out_weights: *mut f32,
                                     out_counts: *mut u32) -> i32;
    fn cqs_hip_bert_embed_submit(e: *mut CqsHipBert, tokens: *const i32, type_ids: *const i32, lens: *const u32, batch: u32,
                                 pooling: u32, ticket: *mut u64) -> i32;
    fn cqs_hip_bert_embed_collect(e: *mut CqsHipBert, ticket: u64, out: *mut f32) -> i32;
    fn cqs_hip_rerank_logits(e: *mut CqsHipBert, tokens: *const i32, type_ids: *const i32, lens: *const u32, batch: u32,
                             out_logits: *mut f32) -> i32;
    fn cqs_hip_bert_embed(e: *mut CqsHipBert, tokens: *const i32, type_ids: *const i32, lens: *const u32, batch: u32,
                          pooling: u32, out: *mut f32) -> i32;
    fn cqs_hip_bert_vocab(e: *const CqsHipBert) -> u32;
    fn cqs_hip_bert_last_error(e: *mut CqsHipBert, buf: *mut c_char, cap: usize) -> usize;
}

pub struct HipBert {
    raw: *mut CqsHipBert,
    cfg: CqsHipBertConfig,
}

// The library serialises calls on one engine with an internal mutex (header, BERT section).
unsafe impl Send for HipBert {}
unsafe impl Sync for HipBert {}

impl HipBert {
    /// `head`: HEAD_MLM (SPLADE) or HEAD_CLASSIFIER (reranker).  `hidden_size` etc. come from the directory's
    /// `config.json` exactly as `probe_splade_hidden_size` reads them today (src/splade/mod.rs:125-150); pass `None`
    /// to keep the preset.
    pub fn open(model_dir: &Path, head: u32, device: i32, overrides: Option<&dyn Fn(&mut CqsHipBertConfig)>) -> Result<Self, String> {
        let mut cfg = unsafe { std::mem::zeroed::<CqsHipBertConfig>() };
        if unsafe { cqs_hip_bert_config_default(head, &mut cfg) } != CQS_HIP_OK {
            return Err("cqs_hip_bert_config_default".into());
        }
        if let Some(f) = overrides {
            f(&mut cfg);
        }
        let dir = CString::new(model_dir.to_string_lossy().as_bytes()).map_err(|e| e.to_string())?;
        let mut raw: *mut CqsHipBert = std::ptr::null_mut();
        let rc = unsafe { cqs_hip_bert_load_dir(dir.as_ptr(), &cfg, device, &mut raw) };
        if rc != CQS_HIP_OK || raw.is_null() {
            return Err(format!("cqs_hip_bert_load_dir({}) failed: {rc}", model_dir.display()));
        }
        Ok(Self { raw, cfg })
    }

    pub fn vocab(&self) -> usize {
        unsafe { cqs_hip_bert_vocab(self.raw) as usize }
    }

    fn last_error(&self) -> String {
        let mut buf = vec![0u8; 512];
        let n = unsafe { cqs_hip_bert_last_error(self.raw, buf.as_mut_ptr() as *mut c_char, buf.len()) };
        String::from_utf8_lossy(&buf[..n]).into_owned()
    }

    fn pack(encodings: &[&[u32]]) -> (Vec<i32>, Vec<u32>) {
        let lens: Vec<u32> = encodings.iter().map(|e| e.len() as u32).collect();
        let toks: Vec<i32> = encodings.iter().flat_map(|e| e.iter().map(|&t| t as i32)).collect();
        (toks, lens)
    }

    /// The pre-pooled `sparse_vector` form of the model output (src/splade/mod.rs:960-978): `[batch, vocab]` f32 of
    /// `ln(1 + max(0, max_s logits))`; the caller keeps `(id, w)` with `w > threshold` as it does today.
    pub fn splade_dense(&self, encodings: &[&[u32]]) -> Result<Vec<f32>, String> {
        let (toks, lens) = Self::pack(encodings);
        let mut out = vec![0f32; encodings.len() * self.vocab()];
        let rc = unsafe { cqs_hip_splade_encode(self.raw, toks.as_ptr(), lens.as_ptr(), lens.len() as u32, out.as_mut_ptr()) };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_splade_encode: {} ({rc})", self.last_error()));
        }
        Ok(out)
    }

    /// `encode_batch` with the threshold filter on the device: `Vec<SparseVector>` directly (ascending ids, weight >
    /// threshold).  A sequence with more than `cap` survivors (never with a trained model: 100-300 entries,
    /// src/splade/mod.rs:44) is re-encoded through `splade_dense`.
    pub fn splade_sparse(&self, encodings: &[&[u32]], threshold: f32, cap: usize) -> Result<Vec<Vec<(u32, f32)>>, String> {
        let (toks, lens) = Self::pack(encodings);
        let b = encodings.len();
        let (mut ids, mut wts, mut cnt) = (vec![0u32; b * cap], vec![0f32; b * cap], vec![0u32; b]);
        let rc = unsafe {
            cqs_hip_splade_encode_sparse(self.raw, toks.as_ptr(), lens.as_ptr(), b as u32, threshold, cap as u32,
                                         ids.as_mut_ptr(), wts.as_mut_ptr(), cnt.as_mut_ptr())
        };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_splade_encode_sparse: {} ({rc})", self.last_error()));
        }
        let mut out = Vec::with_capacity(b);
        for i in 0..b {
            let n = cnt[i] as usize;
            if n <= cap {
                out.push((0..n).map(|j| (ids[i * cap + j], wts[i * cap + j])).collect());
            } else {
                let dense = self.splade_dense(&encodings[i..i + 1])?;
                out.push(dense.iter().enumerate().filter_map(|(id, &v)| if v > threshold { Some((id as u32, v)) } else { None }).collect());
            }
        }
        Ok(out)
    }

    /// Ticket form of `splade_sparse` for the index pipeline (src/splade/mod.rs:774-1075 is called per batch): `submit`
    /// returns once the batch is packed into pinned staging and enqueued on one of the engine's two execution contexts;
    /// keep at most 3 tickets in flight and `collect` (or `abandon`) every one of them - a slot is released only then.
    pub fn splade_submit(&self, encodings: &[&[u32]], threshold: f32, cap: usize) -> Result<SpladeTicket, String> {
        let (toks, lens) = Self::pack(encodings);
        let mut t = 0u64;
        let rc = unsafe {
            cqs_hip_splade_submit_sparse(self.raw, toks.as_ptr(), lens.as_ptr(), lens.len() as u32, threshold, cap as u32, &mut t)
        };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_splade_submit_sparse: {} ({rc})", self.last_error()));
        }
        Ok(SpladeTicket { id: t, batch: encodings.len(), cap, threshold })
    }

    /// Rows with more than `cap` survivors come back as `None`: re-encode those through `splade_dense`.
    pub fn splade_collect(&self, t: SpladeTicket) -> Result<Vec<Option<Vec<(u32, f32)>>>, String> {
        let (b, cap) = (t.batch, t.cap);
        let (mut ids, mut wts, mut cnt) = (vec![0u32; b * cap], vec![0f32; b * cap], vec![0u32; b]);
        let rc = unsafe { cqs_hip_splade_collect_sparse(self.raw, t.id, ids.as_mut_ptr(), wts.as_mut_ptr(), cnt.as_mut_ptr()) };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_splade_collect_sparse: {} ({rc})", self.last_error()));
        }
        Ok((0..b)
            .map(|i| {
                let n = cnt[i] as usize;
                (n <= cap).then(|| (0..n).map(|j| (ids[i * cap + j], wts[i * cap + j])).collect())
            })
            .collect())
    }

    /// Give a ticket's slot back without its results (an `Err` from a later submit, a cancelled index run).
    pub fn splade_abandon(&self, t: SpladeTicket) {
        // SAFETY: NULL buffers are the documented abandon form.
        let _ = unsafe { cqs_hip_splade_collect_sparse(self.raw, t.id, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut()) };
    }

    /// Ticket form of `embed` (the index pipeline's embed stage with a BERT-family preset).
    pub fn embed_submit(&self, encodings: &[&[u32]], type_ids: &[&[u32]], cls: bool) -> Result<(u64, usize), String> {
        let (toks, lens) = Self::pack(encodings);
        let tt: Vec<i32> = type_ids.iter().flat_map(|e| e.iter().map(|&t| t as i32)).collect();
        let mut t = 0u64;
        let rc = unsafe {
            cqs_hip_bert_embed_submit(self.raw, toks.as_ptr(), if tt.is_empty() { std::ptr::null() } else { tt.as_ptr() },
                                      lens.as_ptr(), lens.len() as u32, cls as u32, &mut t)
        };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_bert_embed_submit: {} ({rc})", self.last_error()));
        }
        Ok((t, encodings.len()))
    }

    pub fn embed_collect(&self, ticket: (u64, usize)) -> Result<Vec<f32>, String> {
        let mut out = vec![0f32; ticket.1 * self.cfg.hidden as usize];
        let rc = unsafe { cqs_hip_bert_embed_collect(self.raw, ticket.0, out.as_mut_ptr()) };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_bert_embed_collect: {} ({rc})", self.last_error()));
        }
        Ok(out)
    }

    /// The BERT-family embedder presets (src/embedder/models.rs:346-405): `session.run` + `mean_pool` / `cls_pool`
    /// (src/embedder/pooling.rs:87-128) in one call; `[batch, hidden]`, not normalised - `Embedder::embed_batch` keeps its
    /// `normalize_l2` (core.rs:1196-1203).  `cls`: `PoolingStrategy::Cls`.
    pub fn embed(&self, encodings: &[&[u32]], type_ids: &[&[u32]], cls: bool) -> Result<Vec<f32>, String> {
        let (toks, lens) = Self::pack(encodings);
        let tt: Vec<i32> = type_ids.iter().flat_map(|e| e.iter().map(|&t| t as i32)).collect();
        let mut out = vec![0f32; encodings.len() * self.cfg.hidden as usize];
        let rc = unsafe {
            cqs_hip_bert_embed(self.raw, toks.as_ptr(), if tt.is_empty() { std::ptr::null() } else { tt.as_ptr() },
                               lens.as_ptr(), lens.len() as u32, cls as u32, out.as_mut_ptr())
        };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_bert_embed: {} ({rc})", self.last_error()));
        }
        Ok(out)
    }

    /// `[batch, num_labels]` logits for encoded (query, passage) pairs; `score = sigmoid(logits[i * num_labels])`
    /// (src/reranker.rs:516-518).  `type_ids`: `Encoding::get_type_ids` per pair, or empty when the model takes none.
    pub fn rerank_logits(&self, encodings: &[&[u32]], type_ids: &[&[u32]]) -> Result<Vec<f32>, String> {
        let (toks, lens) = Self::pack(encodings);
        let tt: Vec<i32> = type_ids.iter().flat_map(|e| e.iter().map(|&t| t as i32)).collect();
        let mut out = vec![0f32; encodings.len() * self.cfg.num_labels as usize];
        let rc = unsafe {
            cqs_hip_rerank_logits(self.raw, toks.as_ptr(), if tt.is_empty() { std::ptr::null() } else { tt.as_ptr() },
                                  lens.as_ptr(), lens.len() as u32, out.as_mut_ptr())
        };
        if rc != CQS_HIP_OK {
            return Err(format!("cqs_hip_rerank_logits: {} ({rc})", self.last_error()));
        }
        Ok(out)
    }
}

/// An in-flight `splade_submit` (see there).
pub struct SpladeTicket {
    id: u64,
    batch: usize,
    cap: usize,
    #[allow(dead_code)]
    threshold: f32,
}

impl Drop for HipBert {
    fn drop(&mut self) {
        unsafe { cqs_hip_bert_destroy(self.raw) }
    }
}

// ---- call sites ------------------------------------------------------------------------------------------------
// src/splade/mod.rs, encode_batch, in place of the `session.run` + output match (:900-1075):
//
//     let encs: Vec<&[u32]> = encodings.iter().map(|e| &e.get_ids()[..e.get_ids().len().min(max_seq_len)]).collect();
//     Ok(hip.splade_sparse(&encs, self.threshold, 2048).map_err(SpladeError::InferenceFailed)?)
//     // (or the 2-D `sparse_vector` branch unchanged on `hip.splade_dense(&encs)`: [batch, vocab] activations)
//
// src/reranker.rs, run_chunk, in place of the `session.run` + extraction (:455-520):
//
//     let ids: Vec<&[u32]> = chunk.iter().map(|e| &e.get_ids()[..e.get_ids().len().min(max_len)]).collect();
//     let tys: Vec<&[u32]> = chunk.iter().map(|e| &e.get_type_ids()[..e.get_ids().len().min(max_len)]).collect();
//     let logits = hip.rerank_logits(&ids, &tys).map_err(RerankerError::Inference)?;
//     Ok(logits.chunks(stride).map(|row| Some(sigmoid(row[0]))).collect())
