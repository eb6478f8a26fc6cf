//! Rust binding of `include/kmerhip.h` (C ABI of the MI355X k-mer counter).
//!
//! **Source only** -- never compiled in this repository's build image (no rustc).  It is the
//! binding a krust maintainer would add behind a cargo feature: [`HipKmerMap`] stands where
//! `KmerMap` stands in `src/run.rs:491-583`, [`KmerCounter`] mirrors the fluent builder of
//! `src/builder.rs:95-526` for the packed / string / histogram results.
#![allow(clippy::missing_errors_doc)]

use std::collections::{BTreeMap, HashMap};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

use bytes::Bytes;

pub mod sys {
    use super::{c_char, c_int, c_void};

    #[repr(C)]
    pub struct KhConfig {
        pub struct_size: u32,
        pub k: u32,
        pub min_quality: i32, // -1 = None
        pub device: i32,      // -1 = current
        pub capacity_hint: u64,
        pub stream: *mut c_void,
        pub flags: u32,
        pub input_mib: u32,
    }

    #[repr(C)]
    #[derive(Default, Debug, Clone, Copy)]
    pub struct KhStats {
        pub bases: u64,
        pub kmers: u64,
        pub distinct: u64,
        pub table_slots: u64,
        pub grows: u64,
        pub launches: u64,
        pub count_kernel_ms: f64,
        pub h2d_ms: f64,
        pub part_batches: u64,
        pub stage_ms: [f64; 8],
        pub text_scan_ms: f64,
        pub slot_bytes: u64,
    }

    #[repr(C)]
    pub struct KhCtx {
        _private: [u8; 0],
    }

    /// ncclUniqueId (opaque): names one communicator; rank 0 makes it, every rank gets a copy.
    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct KhUniqueId {
        pub internal: [c_char; 128],
    }

    /// What one rank's `kh_merge_across` did.
    #[repr(C)]
    #[derive(Default, Debug, Clone, Copy)]
    pub struct KhMergeInfo {
        pub route: u32,
        pub pieces: u32,
        pub unit_bytes: u32,
        pub nranks: u32,
        pub local_distinct: u64,
        pub sent_units: u64,
        pub recv_units: u64,
        pub owned_distinct: u64,
        pub export_ms: f64,
        pub wait_ms: f64,
        pub merge_ms: f64,
        pub total_ms: f64,
        pub nranks_seen: u32,
        pub conserved: u32,
        pub sent_count_sum: u64,
        pub merged_count_sum: u64,
    }

    #[repr(C)]
    pub struct KhGroup {
        _private: [u8; 0],
    }

    extern "C" {
        pub fn kh_abi_version() -> c_int;
        pub fn kh_create(out: *mut *mut KhCtx, cfg: *const KhConfig) -> c_int;
        pub fn kh_destroy(ctx: *mut KhCtx);
        pub fn kh_reset(ctx: *mut KhCtx) -> c_int;
        pub fn kh_push(ctx: *mut KhCtx, bases: *const u8, qual: *const u8, n: u64) -> c_int;
        /// format: 1 = FASTA, 2 = FASTQ; -9 (KH_ERR_FORMAT) = parse on the host instead
        pub fn kh_push_text(ctx: *mut KhCtx, text: *const u8, n: u64, format: c_int) -> c_int;
        pub fn kh_finish(ctx: *mut KhCtx, stats: *mut KhStats) -> c_int;
        pub fn kh_result_size(ctx: *mut KhCtx, min_count: u64, n: *mut u64) -> c_int;
        pub fn kh_result_copy(ctx: *mut KhCtx, keys: *mut u64, counts: *mut u64, cap: u64,
                              min_count: u64, n: *mut u64) -> c_int;
        pub fn kh_histogram(ctx: *mut KhCtx, min_count: u64, count: *mut u64, freq: *mut u64,
                            cap: u64, n: *mut u64) -> c_int;
        pub fn kh_lookup(ctx: *mut KhCtx, keys: *const u64, n: u64, counts: *mut u64) -> c_int;
        // host memory the device reaches by DMA (no staging copy in kh_push / kh_push_text / kh_result_copy)
        pub fn kh_host_alloc(out: *mut *mut c_void, bytes: u64) -> c_int;
        pub fn kh_host_free(p: *mut c_void) -> c_int;
        pub fn kh_host_register(p: *mut c_void, bytes: u64) -> c_int;
        pub fn kh_host_unregister(p: *mut c_void) -> c_int;
        pub fn kh_pack(bases: *const u8, k: u32, packed: *mut u64, err_pos: *mut u32) -> c_int;
        pub fn kh_unpack(packed: u64, k: u32, out: *mut u8) -> c_int;
        pub fn kh_canonical(packed: u64, k: u32, canonical: *mut u64, is_rc: *mut c_int) -> c_int;
        pub fn kh_strerror(status: c_int) -> *const c_char;
        pub fn kh_last_error(ctx: *const KhCtx) -> *const c_char;
        // the multi-GPU exchange (RCCL over xGMI) behind the ABI
        pub fn kh_owner(key: u64, k: u32, nparts: u32) -> u32;
        pub fn kh_comm_unique_id(out: *mut KhUniqueId) -> c_int;
        pub fn kh_comm_init(ctx: *mut KhCtx, nranks: u32, rank: u32, id: *const KhUniqueId) -> c_int;
        pub fn kh_merge_across(ctx: *mut KhCtx, info: *mut KhMergeInfo) -> c_int;
        pub fn kh_group_create(out: *mut *mut KhGroup, cfg: *const KhConfig, devices: *const i32, ndevices: u32) -> c_int;
        pub fn kh_group_ctx(g: *mut KhGroup, rank: u32) -> *mut KhCtx;
        pub fn kh_group_size(g: *const KhGroup) -> u32;
        pub fn kh_group_merge(g: *mut KhGroup, infos: *mut KhMergeInfo) -> c_int;
        pub fn kh_group_destroy(g: *mut KhGroup);
    }
}

/// Errors of the device path.  `KmerLength` is `KmerLengthError` of `src/error.rs:86-95`.
#[derive(Debug, thiserror::Error)]
pub enum HipError {
    #[error("k-mer length {k} is out of range: must be between 1 and 32")]
    KmerLength { k: usize },
    #[error("kmerhip: {0}")]
    Device(String),
}

/// `KMERHIP_ABI_VERSION` of the `include/kmerhip.h` that `mod sys` mirrors: the structs above are laid out for it.
pub const ABI_VERSION: c_int = 2;

/// Called before the first `kh_create` of every constructor: a library of another ABI version is refused, not guessed at.
fn check_abi() -> Result<(), HipError> {
    // SAFETY: a pure function of the library
    let v = unsafe { sys::kh_abi_version() };
    if v != ABI_VERSION {
        return Err(HipError::Device(format!("libkmerhip speaks ABI version {v}, this crate {ABI_VERSION}")));
    }
    Ok(())
}

fn check(ctx: *const sys::KhCtx, rc: c_int) -> Result<(), HipError> {
    if rc == 0 {
        return Ok(());
    }
    // SAFETY: both functions return NUL-terminated static / context-owned strings
    let mut msg = unsafe { CStr::from_ptr(sys::kh_strerror(rc)) }.to_string_lossy().into_owned();
    if !ctx.is_null() {
        let detail = unsafe { CStr::from_ptr(sys::kh_last_error(ctx)) }.to_string_lossy();
        if !detail.is_empty() {
            msg = format!("{msg} ({detail})");
        }
    }
    Err(HipError::Device(msg))
}

/// Pinned host memory (`kh_host_alloc`): `kh_push` / `kh_push_text` from it and `kh_result_copy` into it move by DMA,
/// without the staging `memcpy` pageable memory needs.  Where `src/reader.rs:58-79` collects records into
/// `Vec<Bytes>`, a host that wants PCIe speed appends them to one of these instead.
pub struct PinnedBuf {
    ptr: *mut u8,
    cap: usize,
    len: usize,
}

unsafe impl Send for PinnedBuf {}

impl PinnedBuf {
    pub fn with_capacity(cap: usize) -> Result<Self, HipError> {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { sys::kh_host_alloc(&mut p, cap.max(1) as u64) })?;
        Ok(Self { ptr: p as *mut u8, cap, len: 0 })
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    pub fn clear(&mut self) {
        self.len = 0;
    }
    pub fn room(&self) -> usize {
        self.cap - self.len
    }
    /// Appends `s` (the caller checks `room()` first).
    pub fn extend_from_slice(&mut self, s: &[u8]) {
        assert!(s.len() <= self.room());
        // SAFETY: [ptr + len, ptr + len + s.len()) lies inside the allocation and does not overlap `s`
        unsafe { std::ptr::copy_nonoverlapping(s.as_ptr(), self.ptr.add(self.len), s.len()) };
        self.len += s.len();
    }
    pub fn as_slice(&self) -> &[u8] {
        // SAFETY: the first `len` bytes are initialised
        unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
    }
}

impl Drop for PinnedBuf {
    fn drop(&mut self) {
        unsafe { sys::kh_host_free(self.ptr as *mut c_void) };
    }
}

/// Drop-in for `KmerMap` (`src/run.rs:491-583`): `build` / `build_with_quality` / `into_hashmap`.
pub struct HipKmerMap {
    ctx: *mut sys::KhCtx,
    k: usize,
    has_qual: bool,
}

// one producer thread per context (kmerhip.h); moving the handle between threads is fine
unsafe impl Send for HipKmerMap {}

const BATCH: usize = 256 << 20;

impl HipKmerMap {
    pub fn new(k: usize, min_quality: Option<u8>) -> Result<Self, HipError> {
        if !(1..=32).contains(&k) {
            return Err(HipError::KmerLength { k }); // KmerLength::new, src/kmer.rs:100-110
        }
        let cfg = sys::KhConfig {
            struct_size: std::mem::size_of::<sys::KhConfig>() as u32,
            k: k as u32,
            min_quality: min_quality.map_or(-1, i32::from),
            device: -1,
            capacity_hint: 0,
            stream: std::ptr::null_mut(),
            flags: 0,
            input_mib: 0,
        };
        let mut ctx = std::ptr::null_mut();
        check_abi()?;
        check(std::ptr::null(), unsafe { sys::kh_create(&mut ctx, &cfg) })?;
        Ok(Self { ctx, k, has_qual: min_quality.is_some() })
    }

    fn push(&self, bases: &[u8], qual: Option<&[u8]>) -> Result<(), HipError> {
        if bases.is_empty() {
            return Ok(());
        }
        let q = qual.map_or(std::ptr::null(), <[u8]>::as_ptr);
        check(self.ctx, unsafe { sys::kh_push(self.ctx, bases.as_ptr(), q, bases.len() as u64) })
    }

    /// `KmerMap::build` (`src/run.rs:500-503`).  Records are concatenated into flat buffers with a
    /// `\n` between them: any byte outside `ACGTacgt` separates records, no k-mer spans it.
    pub fn build<I: Iterator<Item = Bytes>>(self, sequences: I) -> Result<Self, HipError> {
        // the batch buffer is pinned: kh_push DMAs from it, no staging copy inside the library
        let mut flat = PinnedBuf::with_capacity(BATCH + (1 << 20))?;
        for seq in sequences {
            if seq.len() + 1 > flat.room() {
                self.push(flat.as_slice(), None)?;
                flat.clear();
            }
            if seq.len() + 1 > flat.room() {
                // a record larger than the batch buffer (a chromosome): straight from its own (pageable) memory, then the separator
                self.push(&seq, None)?;
                continue;
            }
            flat.extend_from_slice(&seq);
            flat.extend_from_slice(b"\n");
        }
        self.push(flat.as_slice(), None)?;
        Ok(self)
    }

    /// `KmerMap::build_with_quality` (`src/run.rs:505-520`); the threshold was given to `new`.
    /// A record without qualities (FASTA) is never masked (`run.rs:543`: both must be `Some`).
    pub fn build_with_quality<I>(self, sequences: I) -> Result<Self, HipError>
    where
        I: Iterator<Item = (Bytes, Option<Vec<u8>>)>,
    {
        let (mut b, mut q) = (Vec::with_capacity(BATCH), Vec::with_capacity(BATCH));
        for (seq, qual) in sequences {
            b.extend_from_slice(&seq);
            b.push(b'\n');
            match qual {
                Some(qs) => q.extend_from_slice(&qs),
                None => q.resize(q.len() + seq.len(), 0xFF), // 0xFF >= any threshold (<= 255): never masked
            }
            q.push(b'\n');
            if b.len() >= BATCH {
                self.push(&b, self.has_qual.then_some(&q[..]))?;
                b.clear();
                q.clear();
            }
        }
        self.push(&b, self.has_qual.then_some(&q[..]))?;
        Ok(self)
    }

    /// Raw FASTA / FASTQ text (whole records, uncompressed), records found on the device instead of by
    /// the `bio` readers (`src/reader.rs:58-79`).  `Ok(false)`: the device scanner declined the layout
    /// (wrapped FASTQ, ...) and counted nothing of THIS text; the caller parses the text as today and calls
    /// `build` / `build_with_quality`.  `text` may be reused when the call returns; the counting of an accepted text may
    /// run under the next call's copy (it is finished by `into_packed` / `kh_finish` at the latest).
    pub fn push_text(&mut self, text: &[u8], fastq: bool) -> Result<bool, HipError> {
        let rc = unsafe { sys::kh_push_text(self.ctx, text.as_ptr(), text.len() as u64, if fastq { 2 } else { 1 }) };
        if rc == -9 {
            return Ok(false); // KH_ERR_FORMAT
        }
        check(self.ctx, rc)?;
        Ok(true)
    }

    /// Packed canonical key -> count: the shape of `count_kmers_from_sequences`
    /// (`src/streaming.rs:198-204`).
    pub fn into_packed(self, min_count: u64) -> Result<HashMap<u64, u64>, HipError> {
        check(self.ctx, unsafe { sys::kh_finish(self.ctx, std::ptr::null_mut()) })?;
        let mut n = 0u64;
        check(self.ctx, unsafe { sys::kh_result_size(self.ctx, min_count, &mut n) })?;
        let mut keys = vec![0u64; n as usize];
        let mut counts = vec![0u64; n as usize];
        let mut got = 0u64;
        check(self.ctx, unsafe {
            sys::kh_result_copy(self.ctx, keys.as_mut_ptr(), counts.as_mut_ptr(), n, min_count, &mut got)
        })?;
        Ok(keys.into_iter().zip(counts).take(got as usize).collect())
    }

    /// `KmerMap::into_hashmap` (`src/run.rs:573-582`).
    pub fn into_hashmap(self) -> Result<HashMap<String, u64>, HipError> {
        let k = self.k;
        Ok(self.into_packed(1)?.into_iter().map(|(bits, c)| (unpack_to_string(bits, k), c)).collect())
    }

    /// Count-of-counts after the `min_count` filter, ascending (`src/histogram.rs:88-94`).
    pub fn into_histogram(self, min_count: u64) -> Result<BTreeMap<u64, u64>, HipError> {
        check(self.ctx, unsafe { sys::kh_finish(self.ctx, std::ptr::null_mut()) })?;
        let mut cap = 1u64 << 16;
        loop {
            let (mut c, mut f) = (vec![0u64; cap as usize], vec![0u64; cap as usize]);
            let mut n = 0u64;
            let rc = unsafe { sys::kh_histogram(self.ctx, min_count, c.as_mut_ptr(), f.as_mut_ptr(), cap, &mut n) };
            if rc == -8 {
                cap *= 16; // KH_ERR_RANGE
                continue;
            }
            check(self.ctx, rc)?;
            return Ok(c.into_iter().zip(f).take(n as usize).collect());
        }
    }
}

/// Several GPUs of one node from ONE process (no reference counterpart: the reference's only parallelism is
/// rayon over records, `src/run.rs:500-503`): one device table per GPU, reads dealt out in batches of whole
/// records, then `kh_group_merge` -- the library's RCCL exchange -- turns the N tables into one table sharded
/// by hash range, and the result is the concatenation of the shards.
pub struct HipKmerMapGroup {
    group: *mut sys::KhGroup,
    k: usize,
    next: usize,
}

unsafe impl Send for HipKmerMapGroup {}

impl HipKmerMapGroup {
    /// `devices`: HIP ordinals, one rank each (`&[0, 1, 2, 3, 4, 5, 6, 7]` for a whole MI355X node).
    pub fn new(k: usize, min_quality: Option<u8>, devices: &[i32]) -> Result<Self, HipError> {
        if !(1..=32).contains(&k) {
            return Err(HipError::KmerLength { k });
        }
        let cfg = sys::KhConfig {
            struct_size: std::mem::size_of::<sys::KhConfig>() as u32,
            k: k as u32,
            min_quality: min_quality.map_or(-1, i32::from),
            device: -1,
            capacity_hint: 0,
            stream: std::ptr::null_mut(),
            flags: 0,
            input_mib: 0,
        };
        let mut group = std::ptr::null_mut();
        check_abi()?;
        check(std::ptr::null(), unsafe { sys::kh_group_create(&mut group, &cfg, devices.as_ptr(), devices.len() as u32) })?;
        Ok(Self { group, k, next: 0 })
    }

    fn ctx(&self, rank: usize) -> *mut sys::KhCtx {
        unsafe { sys::kh_group_ctx(self.group, rank as u32) }
    }

    /// `KmerMap::build` over all devices: batches of whole records go to the ranks in turn.
    pub fn build<I: Iterator<Item = Bytes>>(mut self, sequences: I) -> Result<Self, HipError> {
        let n = unsafe { sys::kh_group_size(self.group) } as usize;
        let mut flat = Vec::with_capacity(BATCH / 4 + (1 << 20));
        let mut flush = |flat: &mut Vec<u8>, next: &mut usize| -> Result<(), HipError> {
            if !flat.is_empty() {
                let c = unsafe { sys::kh_group_ctx(self.group, (*next % n) as u32) };
                *next += 1;
                check(c, unsafe { sys::kh_push(c, flat.as_ptr(), std::ptr::null(), flat.len() as u64) })?;
                flat.clear();
            }
            Ok(())
        };
        let mut next = self.next;
        for seq in sequences {
            flat.extend_from_slice(&seq);
            flat.push(b'\n');
            if flat.len() >= BATCH / 4 {
                flush(&mut flat, &mut next)?;
            }
        }
        flush(&mut flat, &mut next)?;
        self.next = next;
        Ok(self)
    }

    /// Merge (collective over the group's ranks, one host thread each inside the library) and collect:
    /// the shards' key sets are disjoint, so their union is the map.
    pub fn into_packed(self, min_count: u64) -> Result<HashMap<u64, u64>, HipError> {
        let n = unsafe { sys::kh_group_size(self.group) } as usize;
        for r in 0..n {
            check(self.ctx(r), unsafe { sys::kh_finish(self.ctx(r), std::ptr::null_mut()) })?;
        }
        check(self.ctx(0), unsafe { sys::kh_group_merge(self.group, std::ptr::null_mut()) })?;
        let mut out = HashMap::new();
        for r in 0..n {
            let c = self.ctx(r);
            let mut cnt = 0u64;
            check(c, unsafe { sys::kh_result_size(c, min_count, &mut cnt) })?;
            let (mut keys, mut counts) = (vec![0u64; cnt as usize], vec![0u64; cnt as usize]);
            let mut got = 0u64;
            check(c, unsafe { sys::kh_result_copy(c, keys.as_mut_ptr(), counts.as_mut_ptr(), cnt, min_count, &mut got) })?;
            out.extend(keys.into_iter().zip(counts).take(got as usize));
        }
        Ok(out)
    }

    /// `KmerMap::into_hashmap` (`src/run.rs:573-582`).
    pub fn into_hashmap(self) -> Result<HashMap<String, u64>, HipError> {
        let k = self.k;
        Ok(self.into_packed(1)?.into_iter().map(|(bits, c)| (unpack_to_string(bits, k), c)).collect())
    }
}

impl Drop for HipKmerMapGroup {
    fn drop(&mut self) {
        // SAFETY: the group came from kh_group_create; its contexts are destroyed with it
        unsafe { sys::kh_group_destroy(self.group) }
    }
}

/// One process per GPU instead (MPI-style hosts): rank 0 calls `comm_unique_id()` and hands the 128 bytes
/// to every rank; each rank then `join`s with its own `HipKmerMap`, counts its share and calls `merge_across`.
pub fn comm_unique_id() -> Result<sys::KhUniqueId, HipError> {
    let mut id = sys::KhUniqueId { internal: [0; 128] };
    check(std::ptr::null(), unsafe { sys::kh_comm_unique_id(&mut id) })?;
    Ok(id)
}

impl HipKmerMap {
    /// Collective: blocks until all `nranks` ranks have joined.
    pub fn join(&mut self, nranks: u32, rank: u32, id: &sys::KhUniqueId) -> Result<(), HipError> {
        check(self.ctx, unsafe { sys::kh_comm_init(self.ctx, nranks, rank, id) })
    }

    /// Collective: afterwards this map holds exactly the keys with `kh_owner(key, k, nranks) == rank`,
    /// counts summed over all ranks.
    pub fn merge_across(&mut self) -> Result<sys::KhMergeInfo, HipError> {
        let mut info = sys::KhMergeInfo::default();
        check(self.ctx, unsafe { sys::kh_merge_across(self.ctx, &mut info) })?;
        Ok(info)
    }
}

impl Drop for HipKmerMap {
    fn drop(&mut self) {
        // SAFETY: ctx came from kh_create and is destroyed exactly once
        unsafe { sys::kh_destroy(self.ctx) }
    }
}

/// `unpack_to_string` (`src/kmer.rs:451-456`) through the library's helper.
#[must_use]
pub fn unpack_to_string(bits: u64, k: usize) -> String {
    let mut out = vec![0u8; k];
    // SAFETY: out holds k bytes; kh_unpack writes exactly k ASCII bytes
    unsafe { sys::kh_unpack(bits, k as u32, out.as_mut_ptr()) };
    String::from_utf8(out).unwrap_or_default()
}

/// The fluent builder of `src/builder.rs:95-526`, device-backed (sequence sources are the caller's:
/// krust keeps its readers, `src/reader.rs`).
#[derive(Debug, Clone, Default)]
pub struct KmerCounter {
    k: Option<usize>,
    min_count: u64,
    min_quality: Option<u8>,
}

impl KmerCounter {
    #[must_use]
    pub fn new() -> Self {
        Self { k: None, min_count: 1, min_quality: None }
    }
    pub fn k(mut self, k: usize) -> Result<Self, HipError> {
        if !(1..=32).contains(&k) {
            return Err(HipError::KmerLength { k });
        }
        self.k = Some(k);
        Ok(self)
    }
    #[must_use]
    pub fn min_count(mut self, n: u64) -> Self {
        self.min_count = n;
        self
    }
    #[must_use]
    pub fn min_quality(mut self, q: u8) -> Self {
        self.min_quality = Some(q);
        self
    }
    /// `count()` over in-memory sequences: `HashMap<String, u64>` filtered by `min_count`.
    pub fn count_sequences<I: Iterator<Item = Bytes>>(&self, seqs: I) -> Result<HashMap<String, u64>, HipError> {
        let k = self.k.ok_or(HipError::KmerLength { k: 0 })?;
        Ok(HipKmerMap::new(k, None)?
            .build(seqs)?
            .into_packed(self.min_count)?
            .into_iter()
            .map(|(bits, c)| (unpack_to_string(bits, k), c))
            .collect())
    }
    /// `histogram()` over in-memory sequences.
    pub fn histogram_sequences<I: Iterator<Item = Bytes>>(&self, seqs: I) -> Result<BTreeMap<u64, u64>, HipError> {
        let k = self.k.ok_or(HipError::KmerLength { k: 0 })?;
        HipKmerMap::new(k, None)?.build(seqs)?.into_histogram(self.min_count)
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    // the reference's own KATs (tests/library_tests.rs:23-33, 55-64; src/streaming.rs:1150-1162)
    #[test]
    fn kats() {
        let c = KmerCounter::new().k(3).unwrap();
        let m = c.count_sequences(vec![Bytes::from_static(b"ACGT")].into_iter()).unwrap();
        assert_eq!(m.get("ACG"), Some(&2));
        let m = c.count_sequences(vec![Bytes::from_static(b"TTT")].into_iter()).unwrap();
        assert_eq!(m.get("AAA"), Some(&1));
        let c4 = KmerCounter::new().k(4).unwrap();
        let m = c4
            .count_sequences(vec![Bytes::from_static(b"AAAA"), Bytes::from_static(b"TTTT")].into_iter())
            .unwrap();
        assert_eq!(m.len(), 1);
        assert_eq!(m.values().next(), Some(&2));
        assert!(KmerCounter::new().k(0).is_err() && KmerCounter::new().k(33).is_err());
    }
}
