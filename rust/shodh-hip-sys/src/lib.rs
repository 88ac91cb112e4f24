//! Reference-shaped wrappers over the C ABI of `libshodh_hip.so` (`include/shodh_hip.h`; raw bindings in `ffi`,
//! generated from the header by `tools/gen_rust_ffi.py`).
//!
//! The types below carry the METHOD SETS of the reference's own types on this path, so that the two field swaps of
//! INTEGRATION.md are the whole integration:
//!   `HipIndex`      <-> `VamanaIndex`      (src/vector_db/vamana.rs:168-1645; exact path of :770-777 / :1167-1188)
//!   `HipSpannIndex` <-> `SpannIndex`       (src/vector_db/spann.rs:574-693, :879-1003)
//!   `HipEmbedder`   <-> `MiniLMEmbedder`   (`impl Embedder`, src/embeddings/mod.rs:52-88; minilm.rs:1122-1376)
//! Not compiled in this repository's image (no Rust toolchain there); the same entry points are exercised through
//! ctypes by tests/.
pub mod ffi;

use anyhow::{anyhow, Result};
use std::ffi::{CStr, CString};
use std::path::Path;

fn check(rc: i32) -> Result<()> {
    if rc == ffi::SHODH_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::shodh_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow!("shodh_hip error {rc}: {msg}"))
}

fn cpath(p: &Path) -> Result<CString> {
    CString::new(p.to_string_lossy().as_bytes()).map_err(|e| anyhow!("path contains NUL: {e}"))
}

/// `VamanaConfig` (vamana.rs:56-90). With `graph_walk` the index answers like the reference WITHOUT `SHODH_VECTOR_EXACT`: it keeps
/// the Vamana graph on the device and `search` is the reference's greedy walk (same graph, same visits, same answers as an index
/// grown by `add_vector`); otherwise the graph parameters are ignored and `search` is the exact scan (`brute_force_search`).
#[derive(Clone, Debug)]
pub struct VamanaConfig {
    pub dimension: usize,
    pub max_degree: usize,
    pub search_list_size: usize,
    pub alpha: f32,
    pub use_mmap: bool,
    pub device: i32,
    pub graph_walk: bool,
}
impl Default for VamanaConfig {
    fn default() -> Self {
        Self { dimension: 384, max_degree: 32, search_list_size: 75, alpha: 1.2, use_mmap: false, device: 0, graph_walk: false }
    }
}

/// Same method set as `VamanaIndex`. `search` returns ids and distances bit-identical to `brute_force_search`.
/// Not carried over: the legacy serde/bincode `save` / `load` / `index_file_exists` trio (vamana.rs:1470-1660), which nothing in
/// the reference calls any more; persistence is `save_to_file` / `load_from_file` (VAMA v1).
pub struct HipIndex {
    h: *mut ffi::shodh_index,
    dim: usize,
    max_degree: usize,
    graph_walk: bool,
    incremental: std::sync::atomic::AtomicUsize,
}
// search may run concurrently on one handle; add / build / delete take the library's own lock
unsafe impl Send for HipIndex {}
unsafe impl Sync for HipIndex {}

pub const MODEL_TOKEN_WINDOW: usize = 128;          // embeddings/chunking.rs (the tokenizer truncation window, minilm.rs:112-117)
pub const REBUILD_THRESHOLD: usize = 10_000;        // vamana.rs:103
pub const REPAIR_THRESHOLD: usize = 1_000;          // vamana.rs:107

impl HipIndex {
    pub fn new(config: VamanaConfig) -> Result<Self> {
        let mut cfg = ffi::shodh_index_cfg::default();
        unsafe { ffi::shodh_index_cfg_default(&mut cfg) };
        cfg.dim = config.dimension as u32; // metric stays NormalizedDotProduct (retrieval.rs:188-193)
        cfg.device = config.device;
        cfg.max_degree = config.max_degree as u32;
        cfg.search_list_size = config.search_list_size as u32;
        cfg.alpha = config.alpha;
        if config.graph_walk { cfg.scan_mode = ffi::SHODH_SCAN_GRAPH as u32; }
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::shodh_index_create(&cfg, &mut h) })?;
        Ok(Self { h, dim: config.dimension, max_degree: config.max_degree, graph_walk: config.graph_walk, incremental: Default::default() })
    }
    /// vamana.rs:175-187: the storage path only tells the reference where to mmap its vectors; rows live in HBM here
    pub fn with_storage_path(config: VamanaConfig, _storage_path: Option<std::path::PathBuf>) -> Result<Self> { Self::new(config) }
    pub fn len(&self) -> usize { unsafe { ffi::shodh_index_len(self.h) as usize } }
    /// graph mode: some `add_vector` since the last build met a walk whose frontier outgrew its array (thousands of equidistant rows). The rows were
    /// added (the call returned Ok, the counters are in step); the graph may differ from the reference's from there on.
    pub fn graph_overflowed(&self) -> bool { unsafe { ffi::shodh_index_graph_overflowed(self.h) != 0 } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }

    /// vamana.rs:200-284 (ids 0..n-1, tombstones cleared)
    pub fn build(&mut self, vectors: Vec<Vec<f32>>) -> Result<()> {
        for v in &vectors {
            if v.len() != self.dim { return Err(anyhow!("Vector dimension mismatch: expected {}, got {}", self.dim, v.len())); }
        }
        let flat: Vec<f32> = vectors.into_iter().flatten().collect();
        self.build_flat(&flat)
    }
    // the library serialises build / add / delete on its own lock, so a shared reference is enough here; the reference's
    // maintenance methods (`auto_rebuild_if_needed(&self)`, vamana.rs:1290) take `&self` for the same reason
    fn build_flat(&self, flat: &[f32]) -> Result<()> {
        check(unsafe { ffi::shodh_index_build(self.h, flat.as_ptr(), (flat.len() / self.dim.max(1)) as u64) })?;
        self.incremental.store(0, std::sync::atomic::Ordering::Release);
        Ok(())
    }
    pub fn rebuild_from_vectors(&mut self, vectors: Vec<Vec<f32>>) -> Result<()> { self.build(vectors) }

    /// vamana.rs:853-974: returns the new vector id (dense, sequential)
    pub fn add_vector(&mut self, vector: Vec<f32>) -> Result<u32> {
        if vector.len() != self.dim { return Err(anyhow!("Vector dimension mismatch: expected {}, got {}", self.dim, vector.len())); }
        let mut id = 0u32;
        check(unsafe { ffi::shodh_index_add(self.h, vector.as_ptr(), 1, &mut id) })?;
        // the vector that seeds an empty index is not an incremental insert (vamana.rs:888-898)
        if id > 0 { self.incremental.fetch_add(1, std::sync::atomic::Ordering::AcqRel); }
        Ok(id)
    }
    /// n sequential `add_vector` calls in one transfer; returns the first id
    pub fn add_vectors(&mut self, flat: &[f32]) -> Result<u32> {
        let mut id = 0u32;
        check(unsafe { ffi::shodh_index_add(self.h, flat.as_ptr(), (flat.len() / self.dim.max(1)) as u64, &mut id) })?;
        let n = flat.len() / self.dim.max(1);
        self.incremental.fetch_add(n - usize::from(n > 0 && id == 0), std::sync::atomic::Ordering::AcqRel);
        Ok(id)
    }

    /// vamana.rs:764-808 under SHODH_VECTOR_EXACT: `(id, distance)` ascending by `(distance.total_cmp, id)`
    pub fn search(&self, query: &[f32], k: usize) -> Result<Vec<(u32, f32)>> {
        if query.len() != self.dim { return Err(anyhow!("Query dimension mismatch: expected {}, got {}", self.dim, query.len())); }
        if k == 0 || self.is_empty() { return Ok(Vec::new()); }
        let (mut ids, mut dist, mut n) = (vec![0u32; k], vec![0f32; k], 0u32);
        check(unsafe { ffi::shodh_index_search(self.h, query.as_ptr(), 1, k as u32, ids.as_mut_ptr(), dist.as_mut_ptr(), &mut n) })?;
        Ok(ids.into_iter().zip(dist).take(n as usize).collect())
    }
    /// `nq` queries in one pass over the corpus (what makes the GPU worth having): rows of `(id, distance)`
    pub fn search_batch(&self, queries: &[f32], k: usize) -> Result<Vec<Vec<(u32, f32)>>> {
        let nq = queries.len() / self.dim.max(1);
        if nq == 0 || k == 0 { return Ok(vec![Vec::new(); nq]); }
        let (mut ids, mut dist, mut n) = (vec![0u32; nq * k], vec![0f32; nq * k], vec![0u32; nq]);
        check(unsafe { ffi::shodh_index_search(self.h, queries.as_ptr(), nq as u32, k as u32, ids.as_mut_ptr(), dist.as_mut_ptr(), n.as_mut_ptr()) })?;
        Ok((0..nq).map(|q| (0..n[q] as usize).map(|i| (ids[q * k + i], dist[q * k + i])).collect()).collect())
    }

    pub fn mark_deleted(&self, id: u32) -> bool {
        let mut ok = 0;
        unsafe { ffi::shodh_index_mark_deleted(self.h, id, &mut ok) };
        ok != 0
    }
    pub fn is_deleted(&self, id: u32) -> bool { unsafe { ffi::shodh_index_is_deleted(self.h, id) == 1 } }
    pub fn deleted_count(&self) -> usize { unsafe { ffi::shodh_index_deleted_count(self.h) as usize } }
    pub fn deletion_ratio(&self) -> f32 { unsafe { ffi::shodh_index_deletion_ratio(self.h) } }
    pub fn needs_compaction(&self) -> bool { unsafe { ffi::shodh_index_needs_compaction(self.h) == 1 } }
    pub fn clear_deleted(&self) { unsafe { ffi::shodh_index_clear_deleted(self.h) }; }

    // Maintenance bookkeeping as the reference keeps it (vamana.rs:976-1232). The exact index has no graph that degrades, so
    // a "repair" re-prunes nothing and recall is 1; the counters and thresholds still behave as callers (IndexHealth,
    // auto_rebuild_index_if_needed) expect, and a rebuild is how tombstoned rows are compacted away.
    pub fn needs_rebuild(&self) -> bool { self.incremental_insert_count() >= REBUILD_THRESHOLD || self.needs_compaction() }
    pub fn needs_repair(&self) -> bool { (REPAIR_THRESHOLD..REBUILD_THRESHOLD).contains(&self.incremental_insert_count()) }
    pub fn incremental_repair(&self) -> Result<usize> {
        // vamana.rs:1033-1115: the last min(inserts, REPAIR_THRESHOLD) nodes are re-pruned once REPAIR_THRESHOLD inserts have accumulated
        // (on the device for a `graph_walk` index; the exact index has no graph: 0 nodes); the counter moves as the reference's does
        let (n, inserts) = (self.len(), self.incremental_insert_count());
        if n == 0 || inserts < REPAIR_THRESHOLD { return Ok(0); }
        let mut repaired = 0u32;
        if self.graph_walk {
            let count = inserts.min(REPAIR_THRESHOLD).min(n);
            check(unsafe { ffi::shodh_index_incremental_repair(self.h, (n - count) as u32, count as u32, &mut repaired) })?;
        }
        self.incremental.store(inserts.saturating_sub(REPAIR_THRESHOLD), std::sync::atomic::Ordering::Relaxed);
        Ok(repaired as usize)
    }

    /// vamana.rs:1194-1208
    pub fn quality_degraded(&self) -> Result<bool> {
        if self.len() < 100 || self.incremental_insert_count() == 0 { return Ok(false); }
        Ok(self.estimate_recall(50, 10)? < 0.85)
    }
    /// `brute_force_search` (vamana.rs:1167-1188) whatever the scan mode
    pub fn brute_force_search(&self, query: &[f32], k: usize) -> Result<Vec<(u32, f32)>> {
        if query.len() != self.dim { return Err(anyhow!("Query dimension {} doesn't match index dimension {}", query.len(), self.dim)); }
        let (mut ids, mut dist, mut cnt) = (vec![0u32; k.max(1)], vec![0f32; k.max(1)], 0u32);
        check(unsafe { ffi::shodh_index_brute_force_search(self.h, query.as_ptr(), 1, k as u32, ids.as_mut_ptr(), dist.as_mut_ptr(), &mut cnt) })?;
        Ok((0..cnt as usize).map(|i| (ids[i], dist[i])).collect())
    }
    /// `estimate_recall` (vamana.rs:1128-1165). The exact index answers with the brute-force scan itself (1.0 by construction); a
    /// `graph_walk` index compares its walk with `brute_force_search` over `sample_size` stored vectors (every `n / sample_size`-th
    /// row here; the reference shuffles with thread_rng).
    pub fn estimate_recall(&self, sample_size: usize, k: usize) -> Result<f32> {
        let n = self.len();
        if n < 2 || !self.graph_walk { return Ok(1.0); }
        let (sample_size, k) = (sample_size.min(n / 2).max(1), k.min(n - 1));
        let mut row = vec![0f32; self.dim];
        let mut total = 0f32;
        for s in 0..sample_size {
            let i = s * (n / sample_size);
            check(unsafe { ffi::shodh_index_extract_rows(self.h, i as u64, 1, row.as_mut_ptr()) })?;
            let ann: std::collections::HashSet<u32> = self.search(&row, k)?.into_iter().map(|(id, _)| id).collect();
            let exact = self.brute_force_search(&row, k)?;
            total += exact.iter().filter(|(id, _)| ann.contains(id)).count() as f32 / k as f32;
        }
        Ok(total / sample_size as f32)
    }
    pub fn auto_maintain(&self) -> Result<String> {
        if self.needs_rebuild() {
            return Ok(if self.auto_rebuild_if_needed()? { "full_rebuild" } else { "rebuild_skipped" }.to_string());
        }
        if self.needs_repair() { return Ok(format!("repaired_{}_nodes", self.incremental_repair()?)); }
        Ok("no_action".to_string())
    }
    /// vamana.rs:1290-1340: rebuild from the live vectors (ids become positional; callers that keep an id mapping rebuild from
    /// their own source of truth instead, retrieval.rs:1629-1656)
    pub fn auto_rebuild_if_needed(&self) -> Result<bool> {
        if !self.needs_rebuild() { return Ok(false); }
        let live: Vec<f32> = self.extract_live_vectors().into_iter().flatten().collect();
        self.build_flat(&live)?;
        Ok(true)
    }
    pub fn is_rebuilding(&self) -> bool { false }
    pub fn incremental_insert_count(&self) -> usize { self.incremental.load(std::sync::atomic::Ordering::Acquire) }
    pub fn reset_incremental_counter(&self) { self.incremental.store(0, std::sync::atomic::Ordering::Release) }

    /// row i bit-for-bit what `add_vector` / `build` received (retrieval.rs:2504-2516)
    pub fn extract_all_vectors(&self) -> Vec<Vec<f32>> {
        let n = self.len();
        let mut flat = vec![0f32; n * self.dim];
        unsafe { ffi::shodh_index_extract_rows(self.h, 0, n as u64, flat.as_mut_ptr()) };
        flat.chunks(self.dim.max(1)).map(|c| c.to_vec()).collect()
    }
    pub fn extract_live_vectors(&self) -> Vec<Vec<f32>> {
        let n = self.len();
        let (mut flat, mut m) = (vec![0f32; n * self.dim], 0u64);
        unsafe { ffi::shodh_index_extract_live_rows(self.h, flat.as_mut_ptr(), std::ptr::null_mut(), n as u64, &mut m) };
        flat.truncate(m as usize * self.dim);
        flat.chunks(self.dim.max(1)).map(|c| c.to_vec()).collect()
    }

    /// the graph of a `graph_walk` index: (degree per node, adjacency lists concatenated in node order, medoid)
    pub fn graph(&self) -> Result<(Vec<u16>, Vec<u32>, u32)> {
        let (n, stride) = (self.len(), self.max_degree + 1);
        let (mut deg, mut nbr, mut medoid) = (vec![0u32; n], vec![0u32; n * stride], 0u32);
        check(unsafe { ffi::shodh_index_get_graph(self.h, deg.as_mut_ptr(), nbr.as_mut_ptr(), stride as u32, &mut medoid) })?;
        let mut flat = Vec::with_capacity(deg.iter().map(|&d| d as usize).sum());
        for (i, &d) in deg.iter().enumerate() { flat.extend_from_slice(&nbr[i * stride..i * stride + d as usize]); }
        Ok((deg.into_iter().map(|d| d as u16).collect(), flat, medoid))
    }

    /// `save_to_file` (vamana_persist.rs:175-284): VAMA v1. A `graph_walk` index writes its graph; the exact index writes empty
    /// adjacency lists (it builds no graph) and the caller keeps `incremental_inserts` at the rebuild threshold, see INTEGRATION.md
    pub fn save_to_file(&self, path: &Path) -> Result<()> {
        let n = self.len();
        let mut flat = vec![0f32; n * self.dim];
        check(unsafe { ffi::shodh_index_extract_rows(self.h, 0, n as u64, flat.as_mut_ptr()) })?;
        let deleted: Vec<u32> = (0..n as u32).filter(|&i| self.is_deleted(i)).collect();
        let p = cpath(path)?;
        if self.graph_walk && n > 0 {
            let (deg, nbr, medoid) = self.graph()?;
            return check(unsafe {
                ffi::shodh_vama_save(p.as_ptr(), flat.as_ptr(), n as u64, self.dim as u32, self.max_degree as u32, medoid, 0,
                                     if deleted.is_empty() { std::ptr::null() } else { deleted.as_ptr() }, deleted.len() as u32,
                                     self.incremental_insert_count() as u64, deg.as_ptr(), nbr.as_ptr())
            });
        }
        // no graph to write: empty adjacency lists, and `incremental_inserts` raised to the rebuild threshold so that a graph-walking reader
        // (the reference without SHODH_VECTOR_EXACT) rebuilds before it walks -- the same rule as persist.py::save_vamana
        let inserts = if n > 1 { self.incremental_insert_count().max(REBUILD_THRESHOLD) } else { self.incremental_insert_count() };
        check(unsafe {
            ffi::shodh_vama_save(p.as_ptr(), flat.as_ptr(), n as u64, self.dim as u32, self.max_degree as u32, 0, 0,
                                 if deleted.is_empty() { std::ptr::null() } else { deleted.as_ptr() }, deleted.len() as u32,
                                 inserts as u64, std::ptr::null(), std::ptr::null())
        })
    }
    /// `load_from_file` (vamana_persist.rs:290-391): vectors and tombstones of a persisted index, straight into HBM
    pub fn load_from_file(path: &Path) -> Result<Self> {
        let p = cpath(path)?;
        let mut info = ffi::shodh_vama_info::default();
        check(unsafe { ffi::shodh_vama_info_read(p.as_ptr(), &mut info) })?;
        let (n, d) = (info.num_vectors as usize, info.dimension as usize);
        let (mut flat, mut deleted) = (vec![0f32; n * d], vec![0u32; info.deleted_count as usize]);
        check(unsafe { ffi::shodh_vama_load(p.as_ptr(), flat.as_mut_ptr(), deleted.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) })?;
        let idx = Self::new(VamanaConfig { dimension: d, max_degree: info.max_degree as usize, ..Default::default() })?;
        if n > 0 { check(unsafe { ffi::shodh_index_build(idx.h, flat.as_ptr(), n as u64) })?; }
        for id in deleted { idx.mark_deleted(id); }
        // an edge-less file's counter is the "rebuild me" sentinel for graph walkers (save_to_file): an exact index does not inherit it
        let inherited = if info.graph_edges == 0 && n > 1 && info.incremental_inserts as usize >= REBUILD_THRESHOLD { 0 } else { info.incremental_inserts as usize };
        idx.incremental.store(inherited, std::sync::atomic::Ordering::Release);
        Ok(idx)
    }
    /// `load_from_file` keeping the file's graph: the index answers by walking it, like the reference that wrote it
    pub fn load_from_file_with_graph(path: &Path) -> Result<Self> {
        let p = cpath(path)?;
        let mut info = ffi::shodh_vama_info::default();
        check(unsafe { ffi::shodh_vama_info_read(p.as_ptr(), &mut info) })?;
        let (n, d) = (info.num_vectors as usize, info.dimension as usize);
        let (mut flat, mut deleted) = (vec![0f32; n * d], vec![0u32; info.deleted_count as usize]);
        let (mut deg16, mut edges) = (vec![0u16; n], vec![0u32; info.graph_edges as usize]);
        check(unsafe { ffi::shodh_vama_load(p.as_ptr(), flat.as_mut_ptr(), deleted.as_mut_ptr(), deg16.as_mut_ptr(), edges.as_mut_ptr()) })?;
        let idx = Self::new(VamanaConfig { dimension: d, max_degree: info.max_degree as usize, graph_walk: true, ..Default::default() })?;
        if n > 0 {
            let stride = info.max_degree as usize + 1;
            let (deg, mut nbr, mut at) = (deg16.iter().map(|&x| x as u32).collect::<Vec<u32>>(), vec![0u32; n * stride], 0usize);
            for (i, &dg) in deg.iter().enumerate() {
                nbr[i * stride..i * stride + dg as usize].copy_from_slice(&edges[at..at + dg as usize]);
                at += dg as usize;
            }
            check(unsafe { ffi::shodh_index_build_with_graph(idx.h, flat.as_ptr(), n as u64, deg.as_ptr(), nbr.as_ptr(), stride as u32, info.medoid) })?;
        }
        for id in deleted { idx.mark_deleted(id); }
        idx.incremental.store(info.incremental_inserts as usize, std::sync::atomic::Ordering::Release);
        Ok(idx)
    }
    pub fn verify_index_file(path: &Path) -> Result<bool> {
        let p = cpath(path)?;
        let mut info = ffi::shodh_vama_info::default();
        Ok(unsafe { ffi::shodh_vama_info_read(p.as_ptr(), &mut info) } == ffi::SHODH_OK)
    }
}
impl Drop for HipIndex {
    fn drop(&mut self) { unsafe { ffi::shodh_index_destroy(self.h) } }
}

/// `SpannIndex` search side (spann.rs:574-693) on a trained state (`load_from_file`, :879-1003, or `set_trained_state`).
pub struct HipSpannIndex { h: *mut ffi::shodh_index, dim: usize, device: i32 }
unsafe impl Send for HipSpannIndex {}
unsafe impl Sync for HipSpannIndex {}
impl HipSpannIndex {
    pub fn new(dimension: usize, num_probes: usize, device: i32) -> Result<Self> {
        let mut cfg = ffi::shodh_index_cfg::default();
        unsafe { ffi::shodh_index_cfg_default(&mut cfg) };
        cfg.dim = dimension as u32;
        cfg.kind = ffi::SHODH_INDEX_IVFPQ as u32;
        cfg.nprobe = num_probes as u32;
        cfg.device = device;
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::shodh_index_create(&cfg, &mut h) })?;
        Ok(Self { h, dim: dimension, device })
    }
    #[allow(clippy::too_many_arguments)]
    pub fn set_trained_state(&mut self, centroids: &[f32], codebook: &[f32], list_off: &[u64], ids: &[u32], codes: &[u8]) -> Result<()> {
        let p = (list_off.len() - 1) as u32;
        let m = (self.dim / 8) as u32;
        check(unsafe { ffi::shodh_index_set_ivfpq(self.h, centroids.as_ptr(), p, codebook.as_ptr(), m, 256, list_off.as_ptr(), ids.as_ptr(), codes.as_ptr()) })
    }
    pub fn load_from_file(path: &Path, num_probes: usize, device: i32) -> Result<Self> {
        let p = cpath(path)?;
        let mut info = ffi::shodh_span_info::default();
        check(unsafe { ffi::shodh_span_info_read(p.as_ptr(), &mut info) })?;
        if info.pq_enabled != 1 { return Err(anyhow!("PQ-less SPANN files carry no codes to scan")); }
        let (pn, d, m, t) = (info.num_partitions as usize, info.dimension as usize, info.pq_subvectors as usize, info.total_postings as usize);
        let mut cent = vec![0f32; pn * d];
        let mut cb = vec![0f32; m * info.pq_num_centroids as usize * info.pq_subvec_dim as usize];
        let (mut off, mut ids, mut codes) = (vec![0u64; pn + 1], vec![0u32; t], vec![0u8; t * m]);
        check(unsafe { ffi::shodh_span_load(p.as_ptr(), cent.as_mut_ptr(), cb.as_mut_ptr(), off.as_mut_ptr(), ids.as_mut_ptr(), codes.as_mut_ptr()) })?;
        let mut idx = Self::new(d, num_probes, device)?;
        idx.set_trained_state(&cent, &cb, &off, &ids, &codes)?;
        Ok(idx)
    }
    /// `SpannIndex::build` (spann.rs:363-463): ceil(sqrt(n)) partitions, 25 IVF + 20 PQ Lloyd iterations on the device
    /// (bit-identical to the reference's arithmetic given the shuffles), then every vector assigned and PQ-encoded into its
    /// posting list. `shuffle(seed_stream, n)` supplies the initial permutations the reference draws from `thread_rng`
    /// (one for the IVF centroids, then one per PQ sub-space), e.g. `|_, n| { let mut v: Vec<u32> = (0..n as u32).collect();
    /// v.shuffle(&mut rand::thread_rng()); v }`.
    pub fn build(&mut self, vectors: &[Vec<f32>], mut shuffle: impl FnMut(usize, usize) -> Vec<u32>) -> Result<()> {
        let n = vectors.len();
        if n == 0 { return Err(anyhow!("Cannot build index from empty vectors")); }       // spann.rs:364-366
        let (d, m) = (self.dim, self.dim / 8);
        let p = ((n as f64).sqrt().ceil() as usize).max(1);                                 // compute_partitions, spann.rs:135-139
        let flat: Vec<f32> = vectors.iter().flatten().copied().collect();
        let ivf_perm = shuffle(0, n);
        let pq_perms: Vec<u32> = (0..m).flat_map(|s| shuffle(1 + s, n)).collect();
        let (mut cent, mut cb) = (vec![0f32; p * d], vec![0f32; m * 256 * 8]);
        check(unsafe { ffi::shodh_ivfpq_train(self.device, flat.as_ptr(), n as u64, d as u32, p as u32, 25, 20, ivf_perm.as_ptr(), pq_perms.as_ptr(), cent.as_mut_ptr(), cb.as_mut_ptr()) })?;
        self.set_trained_state(&cent, &cb, &vec![0u64; p + 1], &[], &[])?;
        let (mut assign, mut codes) = (vec![0u32; n], vec![0u8; n * m]);
        check(unsafe { ffi::shodh_index_ivfpq_encode(self.h, flat.as_ptr(), n as u64, assign.as_mut_ptr(), codes.as_mut_ptr()) })?;
        // posting lists in insertion order inside a partition (stable counting sort by partition)
        let mut off = vec![0u64; p + 1];
        for &a in &assign { off[a as usize + 1] += 1; }
        for i in 0..p { off[i + 1] += off[i]; }
        let mut cursor = off.clone();
        let (mut ids, mut sorted) = (vec![0u32; n], vec![0u8; n * m]);
        for (i, &a) in assign.iter().enumerate() {
            let at = cursor[a as usize] as usize;
            cursor[a as usize] += 1;
            ids[at] = i as u32;
            sorted[at * m..(at + 1) * m].copy_from_slice(&codes[i * m..(i + 1) * m]);
        }
        self.set_trained_state(&cent, &cb, &off, &ids, &sorted)
    }
    pub fn len(&self) -> usize { unsafe { ffi::shodh_index_len(self.h) as usize } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
    pub fn insert(&mut self, vector_id: u32, vector: &[f32]) -> Result<()> {
        check(unsafe { ffi::shodh_index_ivfpq_insert(self.h, vector_id, vector.as_ptr()) })
    }
    pub fn search(&self, query: &[f32], k: usize) -> Result<Vec<(u32, f32)>> {
        if query.len() != self.dim { return Err(anyhow!("Query dimension mismatch: expected {}, got {}", self.dim, query.len())); }
        if k == 0 { return Ok(Vec::new()); }
        let (mut ids, mut dist, mut n) = (vec![0u32; k], vec![0f32; k], 0u32);
        check(unsafe { ffi::shodh_index_search(self.h, query.as_ptr(), 1, k as u32, ids.as_mut_ptr(), dist.as_mut_ptr(), &mut n) })?;
        Ok(ids.into_iter().zip(dist).take(n as usize).collect())
    }
}
impl Drop for HipSpannIndex {
    fn drop(&mut self) { unsafe { ffi::shodh_index_destroy(self.h) } }
}

/// What the embedder needs from a tokenizer (the reference uses the `tokenizers` crate with truncation 128 and pads to
/// 256, minilm.rs:112-117, :153-154): token ids and attention mask of `text`, special tokens included.
pub trait Tokenize: Send + Sync {
    fn encode(&self, text: &str) -> Result<(Vec<u32>, Vec<u32>)>;
    /// full sequence length without truncation (`count_tokens`, minilm.rs:1216-1229)
    fn count(&self, text: &str) -> usize;
}

/// `impl Embedder` (src/embeddings/mod.rs:52-88). Tokenisation stays on the host; `session.run` + pooling
/// (minilm.rs:939-981) run on the device. Add `impl crate::embeddings::Embedder for HipEmbedder<T>` in the reference
/// tree forwarding to these inherent methods (the trait lives in that crate).
/// GEMM operand type of the encoder. `Int8` is the reference's default model (SHODH_USE_QUANTIZED_MODEL unset, minilm.rs:205-220).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum EncoderDtype { Fp32, Bf16, Int8 }
pub struct HipEmbedder<T: Tokenize> { h: *mut ffi::shodh_embedder, tok: T, dim: usize, max_len: usize }
unsafe impl<T: Tokenize> Send for HipEmbedder<T> {}
unsafe impl<T: Tokenize> Sync for HipEmbedder<T> {}
impl<T: Tokenize> HipEmbedder<T> {
    /// `weights`: the f32 blob in HF `BertModel` parameter order (embedder.py::state_dict_to_blob documents every slice). With
    /// `EncoderDtype::Int8` a blob is quantised by the library (per tensor, symmetric); to run the reference's own quantised
    /// export use `from_model_file`.
    pub fn new(tok: T, weights: &[f32], device: i32, dtype: EncoderDtype) -> Result<Self> {
        let e = Self::create(tok, device, dtype, None)?;
        check(unsafe { ffi::shodh_embedder_load_weights(e.h, weights.as_ptr(), weights.len() as u64) })?;
        Ok(e)
    }
    /// The model file the reference hands to ONNX Runtime (`EmbeddingConfig.model_path`, minilm.rs:212-220): `model_quantized.onnx`
    /// (= onnx/model_quint8_avx2.onnx, the default: its uint8 tensors, scales and zero points are multiplied as they are under
    /// `EncoderDtype::Int8`), `model.onnx`, or the checkpoint's `model.safetensors`.
    pub fn from_model_file(tok: T, model_path: &std::path::Path, device: i32, dtype: EncoderDtype) -> Result<Self> {
        let p = std::ffi::CString::new(model_path.to_string_lossy().as_bytes()).map_err(|_| anyhow!("model path contains NUL"))?;
        Self::create(tok, device, dtype, Some(&p))
    }
    fn create(tok: T, device: i32, dtype: EncoderDtype, path: Option<&std::ffi::CString>) -> Result<Self> {
        let mut cfg = ffi::shodh_embed_cfg::default();
        unsafe { ffi::shodh_embed_cfg_default(&mut cfg) };
        cfg.device = device;
        cfg.dtype = match dtype { EncoderDtype::Fp32 => ffi::SHODH_DTYPE_FP32, EncoderDtype::Bf16 => ffi::SHODH_DTYPE_BF16, EncoderDtype::Int8 => ffi::SHODH_DTYPE_INT8 } as u32;
        // the reference's INT8 tensor is padded to max_length and DynamicQuantizeLinear ranges span it (minilm.rs:588-593)
        cfg.compute_padded = if dtype == EncoderDtype::Int8 { 1 } else { 0 };
        cfg.weights_path = path.map_or(std::ptr::null(), |p| p.as_ptr());
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::shodh_embedder_create(&cfg, &mut h) })?;
        Ok(Self { h, tok, dim: cfg.hidden as usize, max_len: cfg.max_len as usize })
    }
    pub fn dimension(&self) -> usize { self.dim }
    pub fn encode(&self, text: &str) -> Result<Vec<f32>> {
        if text.is_empty() { return Ok(vec![0.0; self.dim]); } // minilm.rs:1123-1125
        Ok(self.encode_batch(&[text])?.pop().unwrap())
    }
    pub fn encode_query(&self, text: &str) -> Result<Vec<f32>> { self.encode(text) } // symmetric model
    /// minilm.rs:1247-1376: one device call for the whole batch; empty texts give zero vectors. INT8: `SHODH_QUANT_SCOPE_BATCH`, the
    /// reference's own `encode_batch` arithmetic (one `session.run` on `[B, max_len]`, DynamicQuantizeLinear ranges over the batch tensor).
    pub fn encode_batch(&self, texts: &[&str]) -> Result<Vec<Vec<f32>>> { self.encode_scoped(texts, ffi::SHODH_QUANT_SCOPE_BATCH as u32) }
    /// N x `encode()` in one device call: what `remember` / `index_memory` / `recall` compute text by text (memory/mod.rs:1037,
    /// retrieval.rs:673, :708, :878 -- one `session.run` on `[1, max_len]` each, minilm.rs:883-982). INT8: `SHODH_QUANT_SCOPE_PER_TEXT`,
    /// every range spans one text's padded tensor; bit-identical to calling `encode` N times. fp32 / bf16: the same as `encode_batch`.
    pub fn encode_each(&self, texts: &[&str]) -> Result<Vec<Vec<f32>>> { self.encode_scoped(texts, ffi::SHODH_QUANT_SCOPE_PER_TEXT as u32) }
    fn encode_scoped(&self, texts: &[&str], scope: u32) -> Result<Vec<Vec<f32>>> {
        // the scope travels with the call (ABI v5): concurrent encode / encode_each / encode_batch callers cannot disturb each other, and
        // concurrent one-text calls are coalesced by the library into one per-text forward (same bytes per text)
        let b = texts.len();
        let (mut ids, mut mask) = (vec![0i32; b * self.max_len], vec![0u8; b * self.max_len]);
        for (r, text) in texts.iter().enumerate() {
            if text.is_empty() { continue; }
            let (t, m) = self.tok.encode(text)?;
            for (i, (&tt, &mm)) in t.iter().zip(m.iter()).take(self.max_len).enumerate() {
                ids[r * self.max_len + i] = tt as i32;
                mask[r * self.max_len + i] = mm as u8;
            }
        }
        let mut out = vec![0f32; b * self.dim];
        check(unsafe { ffi::shodh_embedder_encode_ids_scoped(self.h, ids.as_ptr(), mask.as_ptr(), b as u32, scope, out.as_mut_ptr()) })?;
        Ok(out.chunks(self.dim.max(1)).map(|c| c.to_vec()).collect())
    }
    pub fn count_tokens(&self, text: &str) -> usize { self.tok.count(text) }
    /// minilm.rs:1236-1245 for a model without a document prefix (MiniLM is symmetric): the whole window
    pub fn chunk_budget_tokens(&self) -> usize { MODEL_TOKEN_WINDOW }
}
impl<T: Tokenize> Drop for HipEmbedder<T> {
    fn drop(&mut self) { unsafe { ffi::shodh_embedder_destroy(self.h) } }
}

/// Data-parallel encoder over several devices (texts are independent, no collective): one `HipEmbedder` per GPU, `encode_each` deals the
/// texts to them in contiguous near-equal shares, one host thread per share, vectors back in input order -- byte for byte what ONE handle returns
/// (the per-text scope: a text's vector does not depend on its batch mates). `encode` / `encode_query` / `encode_batch` go to handle 0: an INT8
/// `encode_batch` is one function of the whole batch (its activation ranges span it, minilm.rs:588-593) and must not be split. The Python mirror
/// (`shodh_memory_amd.distributed.ShardedEmbedder`) is tested with several handles on one GPU.
pub struct HipShardedEmbedder<T: Tokenize> { pub handles: Vec<HipEmbedder<T>> }
impl<T: Tokenize> HipShardedEmbedder<T> {
    pub fn new(handles: Vec<HipEmbedder<T>>) -> Result<Self> {
        if handles.is_empty() { return Err(anyhow!("at least one embedder handle")); }
        Ok(Self { handles })
    }
    pub fn dimension(&self) -> usize { self.handles[0].dimension() }
    pub fn encode(&self, text: &str) -> Result<Vec<f32>> { self.handles[0].encode(text) }
    pub fn encode_query(&self, text: &str) -> Result<Vec<f32>> { self.handles[0].encode_query(text) }
    pub fn encode_batch(&self, texts: &[&str]) -> Result<Vec<Vec<f32>>> { self.handles[0].encode_batch(texts) }
    pub fn encode_each(&self, texts: &[&str]) -> Result<Vec<Vec<f32>>> {
        let g = self.handles.len().min(texts.len().max(1));
        let per = (texts.len() + g - 1) / g.max(1);
        let parts: Vec<Result<Vec<Vec<f32>>>> = std::thread::scope(|sc| {
            let joins: Vec<_> = texts.chunks(per.max(1)).zip(self.handles.iter()).map(|(share, h)| sc.spawn(move || h.encode_each(share))).collect();
            joins.into_iter().map(|j| j.join().unwrap_or_else(|_| Err(anyhow!("encoder thread panicked")))).collect()
        });
        let mut out = Vec::with_capacity(texts.len());
        for p in parts { out.extend(p?); }
        Ok(out)
    }
    pub fn count_tokens(&self, text: &str) -> usize { self.handles[0].count_tokens(text) }
    pub fn chunk_budget_tokens(&self) -> usize { self.handles[0].chunk_budget_tokens() }
}

/// `LearnedWeights::fuse_scores_full` (src/relevance.rs:529-594) with a `#[repr(C)]` copy of the seven weights
pub fn fuse_scores_full(w: &ffi::shodh_weights, sem: f32, ent: f32, tag: f32, imp: f32, momentum_ema: f32, access_count: u32, graph_strength: f32) -> f32 {
    unsafe { ffi::shodh_fuse_scores_full(w, sem, ent, tag, imp, momentum_ema, access_count, graph_strength) }
}

/// Recall Layer 4 (src/memory/mod.rs:3878-4468): the hybrid (vector + BM25) leg and the graph leg fused into one score per
/// memory. `hybrid`: `(id, bm25_score, vector_score)` in hybrid rank order; `graph`: `(id, activation)` in rank order (ids are
/// the 16 uuid bytes of `MemoryId`). `cfg` comes from `shodh_leg_fusion_cfg_default` with the SHODH_* switches recall reads
/// from the environment copied into its fields and `graph_w` / `hybrid_w` from `shodh_leg_fusion_weights`.
/// Returns the fused map (sorted score desc, id asc) and `effective_vec_trust`.
pub fn fuse_legs(cfg: &ffi::shodh_leg_fusion_cfg, hybrid: &[([u8; 16], f32, f32)], graph: &[([u8; 16], f32)], query_len: usize) -> (Vec<([u8; 16], f32)>, f32) {
    let hu: Vec<u8> = hybrid.iter().flat_map(|h| h.0).collect();
    let hb: Vec<f32> = hybrid.iter().map(|h| h.1).collect();
    let hv: Vec<f32> = hybrid.iter().map(|h| h.2).collect();
    let gu: Vec<u8> = graph.iter().flat_map(|g| g.0).collect();
    let ga: Vec<f32> = graph.iter().map(|g| g.1).collect();
    let cap = hybrid.len() + graph.len();
    let mut ou = vec![0u8; cap.max(1) * 16];
    let mut os = vec![0f32; cap.max(1)];
    let mut trust = 1.0f32;
    let n = unsafe {
        ffi::shodh_fuse_legs(cfg, hu.as_ptr(), hb.as_ptr(), hv.as_ptr(), hybrid.len(), gu.as_ptr(), ga.as_ptr(), graph.len(), query_len,
                             ou.as_mut_ptr(), os.as_mut_ptr(), cap, &mut trust)
    };
    let out = (0..n.min(cap)).map(|i| { let mut u = [0u8; 16]; u.copy_from_slice(&ou[i * 16..i * 16 + 16]); (u, os[i]) }).collect();
    (out, trust)
}

/// ONE `VamanaIndex`-shaped index over several GPUs of a node (BASELINE.json configs[4]): per-device shards,
/// RCCL all-gather of the per-shard top-k, merge with the comparator of vamana.rs:1185 -- all inside the library
/// (`shodh_sharded_index_*`, csrc/sharded.hip). Same ids, same results as a `HipIndex` holding the whole corpus.
pub struct HipShardedIndex {
    h: *mut ffi::shodh_sharded_index,
    dim: usize,
}
unsafe impl Send for HipShardedIndex {}
unsafe impl Sync for HipShardedIndex {}

impl HipShardedIndex {
    /// `devices`: HIP ordinals, one shard each (e.g. `&[0, 1, 2, 3, 4, 5, 6, 7]`)
    pub fn new(config: VamanaConfig, devices: &[i32]) -> Result<Self> {
        let mut cfg = ffi::shodh_sharded_cfg::default();
        unsafe { ffi::shodh_sharded_cfg_default(&mut cfg) };
        cfg.dim = config.dimension as u32;
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::shodh_sharded_index_create(&cfg, devices.as_ptr(), devices.len() as u32, &mut h) })?;
        Ok(Self { h, dim: config.dimension })
    }
    pub fn len(&self) -> usize { unsafe { ffi::shodh_sharded_index_len(self.h) as usize } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
    pub fn shards(&self) -> usize { unsafe { ffi::shodh_sharded_index_shards(self.h) as usize } }
    pub fn uses_rccl(&self) -> bool { unsafe { ffi::shodh_sharded_index_uses_rccl(self.h) == 1 } }
    pub fn build(&mut self, vectors: Vec<Vec<f32>>) -> Result<()> {
        for v in &vectors {
            if v.len() != self.dim { return Err(anyhow!("Vector dimension mismatch: expected {}, got {}", self.dim, v.len())); }
        }
        let flat: Vec<f32> = vectors.into_iter().flatten().collect();
        check(unsafe { ffi::shodh_sharded_index_build(self.h, flat.as_ptr(), (flat.len() / self.dim.max(1)) as u64) })
    }
    pub fn add_vector(&mut self, vector: Vec<f32>) -> Result<u32> {
        if vector.len() != self.dim { return Err(anyhow!("Vector dimension mismatch: expected {}, got {}", self.dim, vector.len())); }
        let mut id = 0u32;
        check(unsafe { ffi::shodh_sharded_index_add(self.h, vector.as_ptr(), 1, &mut id) })?;
        Ok(id)
    }
    pub fn search(&self, query: &[f32], k: usize) -> Result<Vec<(u32, f32)>> {
        if query.len() != self.dim { return Err(anyhow!("Query dimension mismatch: expected {}, got {}", self.dim, query.len())); }
        if k == 0 || self.is_empty() { return Ok(Vec::new()); }
        let (mut ids, mut dist, mut n) = (vec![0u32; k], vec![0f32; k], 0u32);
        check(unsafe { ffi::shodh_sharded_index_search(self.h, query.as_ptr(), 1, k as u32, ids.as_mut_ptr(), dist.as_mut_ptr(), &mut n) })?;
        Ok(ids.into_iter().zip(dist).take(n as usize).collect())
    }
    pub fn search_batch(&self, queries: &[f32], k: usize) -> Result<Vec<Vec<(u32, f32)>>> {
        let nq = queries.len() / self.dim.max(1);
        if nq == 0 || k == 0 { return Ok(vec![Vec::new(); nq]); }
        let (mut ids, mut dist, mut n) = (vec![0u32; nq * k], vec![0f32; nq * k], vec![0u32; nq]);
        check(unsafe { ffi::shodh_sharded_index_search(self.h, queries.as_ptr(), nq as u32, k as u32, ids.as_mut_ptr(), dist.as_mut_ptr(), n.as_mut_ptr()) })?;
        Ok((0..nq).map(|q| (0..n[q] as usize).map(|i| (ids[q * k + i], dist[q * k + i])).collect()).collect())
    }
    pub fn mark_deleted(&self, id: u32) -> bool {
        let mut ok = 0;
        unsafe { ffi::shodh_sharded_index_mark_deleted(self.h, id, &mut ok) };
        ok != 0
    }
    pub fn is_deleted(&self, id: u32) -> bool { unsafe { ffi::shodh_sharded_index_is_deleted(self.h, id) == 1 } }
    pub fn deleted_count(&self) -> usize { unsafe { ffi::shodh_sharded_index_deleted_count(self.h) as usize } }
    pub fn clear_deleted(&self) { unsafe { ffi::shodh_sharded_index_clear_deleted(self.h) }; }
    pub fn extract_all_vectors(&self) -> Result<Vec<Vec<f32>>> {
        let n = self.len();
        let mut flat = vec![0f32; n * self.dim];
        check(unsafe { ffi::shodh_sharded_index_extract_rows(self.h, 0, n as u64, flat.as_mut_ptr()) })?;
        Ok(flat.chunks(self.dim.max(1)).map(|c| c.to_vec()).collect())
    }
}
impl Drop for HipShardedIndex {
    fn drop(&mut self) { unsafe { ffi::shodh_sharded_index_destroy(self.h) } }
}
