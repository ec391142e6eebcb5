//! `msbwt2-hip`: `GpuRleBWT`, an MI355X-backed drop-in for `msbwt2::rle_bwt::RleBWT`, over the C ABI of
//! `libmsbwt_hip.so` (include/msbwt_hip.h of the rust-msbwt_amd repository).  It implements
//! `msbwt2::msbwt_core::BWT` (src/msbwt_core.rs:28-162 of rust-msbwt), so code written against the
//! trait runs unchanged; the batch methods (`count_kmers`, `constrain_ranges`, `count_read_kmers`) are
//! where the GPU pays off.
//!
//! NOT COMPILED in the repository's build image (no cargo/rustc there): written against the header;
//! tests/test_shim_matches_header.py keeps every `extern "C"` declaration below in step with it.
use std::ffi::{c_char, c_int, c_void, CStr, CString};
use std::io;

use msbwt2::msbwt_core::{BWTRange, BWT, VC_LEN};

#[repr(C)]
pub struct MsbwtRle { _private: [u8; 0] }

const MSBWT_OK: c_int = 0;
const MSBWT_ERR_IO: c_int = -1;
const MSBWT_ERR_UNEXPECTED_EOF: c_int = -2;

extern "C" {
    fn msbwt_rle_new(bin_power: u8) -> *mut MsbwtRle;
    fn msbwt_rle_new_on_device(bin_power: u8, device: c_int) -> *mut MsbwtRle;
    fn msbwt_rle_free(bwt: *mut MsbwtRle);
    fn msbwt_rle_load_vector(bwt: *mut MsbwtRle, rle: *const u8, len: usize) -> c_int;
    fn msbwt_rle_load_numpy_file(bwt: *mut MsbwtRle, path: *const c_char) -> c_int;
    fn msbwt_rle_get_symbol_count(bwt: *const MsbwtRle, symbol: u8) -> u64;
    fn msbwt_rle_get_total_size(bwt: *const MsbwtRle) -> u64;
    fn msbwt_rle_constrain_range(bwt: *const MsbwtRle, sym: u8, l: u64, h: u64,
                                 out_l: *mut u64, out_h: *mut u64) -> c_int;
    fn msbwt_rle_count_kmer(bwt: *const MsbwtRle, kmer: *const u8, k: usize, out: *mut u64) -> c_int;
    fn msbwt_rle_count_kmers(bwt: *const MsbwtRle, kmers: *const u8, k: usize, n: usize,
                             out_counts: *mut u64) -> c_int;
    fn msbwt_rle_constrain_ranges(bwt: *const MsbwtRle, syms: *const u8, l: *const u64, h: *const u64,
                                  n: usize, out_l: *mut u64, out_h: *mut u64) -> c_int;
    fn msbwt_rle_count_read_kmers(bwt: *const MsbwtRle, reads: *const u8, read_len: usize, n_reads: usize,
                                  k: usize, ascii: c_int, out_fwd: *mut u64, out_rc: *mut u64) -> c_int;
    fn msbwt_rle_count_kmers_device(bwt: *const MsbwtRle, d_kmers: *const c_void, k: usize, n: usize,
                                    d_out: *mut c_void, hip_stream: *mut c_void) -> c_int;
    fn msbwt_rle_device_status(bwt: *const MsbwtRle, hip_stream: *mut c_void) -> c_int;
    fn msbwt_rle_set_table_depth(bwt: *mut MsbwtRle, depth: c_int) -> c_int;
    fn msbwt_rle_last_error(bwt: *const MsbwtRle) -> *const c_char;
    // several GPUs of one node
    fn msbwt_rle_replicate(src: *const MsbwtRle, device: c_int) -> *mut MsbwtRle;
    fn msbwt_rle_count_kmers_multi(replicas: *const *const MsbwtRle, n_replicas: usize, kmers: *const u8,
                                   k: usize, n: usize, out_counts: *mut u64) -> c_int;
    fn msbwt_rle_count_read_kmers_multi(replicas: *const *const MsbwtRle, n_replicas: usize, reads: *const u8,
                                        read_len: usize, n_reads: usize, k: usize, ascii: c_int,
                                        out_fwd: *mut u64, out_rc: *mut u64) -> c_int;
    // one process per GPU: the final count gather over RCCL (librccl.so is bound at run time)
    fn msbwt_comm_get_unique_id(out_id: *mut c_void) -> c_int;
    fn msbwt_comm_init_rank(out_comm: *mut *mut c_void, nranks: c_int, id: *const c_void, rank: c_int) -> c_int;
    fn msbwt_comm_destroy(comm: *mut c_void) -> c_int;
    fn msbwt_rle_allgather_counts(bwt: *const MsbwtRle, comm: *mut c_void, d_mine: *const c_void, n_mine: usize,
                                  d_all: *mut c_void, wire_bits: c_int, hip_stream: *mut c_void) -> c_int;
    // batch order: sort a batch by these keys (ascending) and it walks the index in order; the library's own ordering pass is a
    // switch (1 = whenever the pass applies; 0 and -1, the default, = never: there is no automatic mode)
    fn msbwt_kmer_order_keys(kmers: *const u8, k: usize, n: usize, out_keys: *mut u64) -> c_int;
    fn msbwt_rle_set_batch_order(bwt: *mut MsbwtRle, mode: c_int) -> c_int;
    // compact queries: two bits per symbol -- 8 bytes per 31-mer over PCIe instead of 31, 32-bit counts on the way back
    fn msbwt_kmers_pack_2bit(kmers: *const u8, k: usize, n: usize, out_words: *mut u64) -> c_int;
    fn msbwt_rle_count_kmers_packed(bwt: *const MsbwtRle, kmers2bit: *const u64, k: usize, n: usize,
                                    out_counts: *mut c_void, count_bits: c_int) -> c_int;
    // HBM the index may hold (0 = no budget): the space / time knob, as bin_power is the reference's
    fn msbwt_rle_set_memory_budget(bwt: *mut MsbwtRle, bytes: u64) -> c_int;
    // sparse suffix table (round 5): ranges of the suffixes that occur, depth 16..31 (-1 = automatic, 0 = off); info = 80 u64 words
    fn msbwt_rle_set_sparse_table(bwt: *mut MsbwtRle, depth: c_int) -> c_int;
    fn msbwt_rle_get_sparse_table(bwt: *const MsbwtRle) -> c_int;
    // the k the index will mostly be asked about (0 = unknown): the automatic sparse table then goes as deep as min(k, 31) where its table fits, instead of 23
    fn msbwt_rle_set_query_length(bwt: *mut MsbwtRle, k: c_int) -> c_int;
    fn msbwt_rle_get_query_length(bwt: *const MsbwtRle) -> c_int;
    fn msbwt_rle_set_sparse_tiers(bwt: *mut MsbwtRle, mode: c_int) -> c_int;
    fn msbwt_rle_get_sparse_tiers(bwt: *const MsbwtRle) -> c_int;
    fn msbwt_rle_set_sparse_second(bwt: *mut MsbwtRle, mode: c_int) -> c_int;
    fn msbwt_auto_sparse_max_depth(query_length: c_int) -> c_int;
    fn msbwt_rle_sparse_table_info(bwt: *const MsbwtRle, out: *mut u64) -> c_int;
    // one batch counted and gathered as a pipeline (pieces searched while earlier pieces' counts travel over RCCL)
    fn msbwt_rle_count_kmers_allgather_device(bwt: *const MsbwtRle, comm: *mut c_void, d_kmers: *const c_void, k: usize, n_mine: usize,
                                              d_mine_counts: *mut c_void, d_all: *mut c_void, wire_bits: c_int, out_bits: c_int,
                                              pieces: c_int, hip_stream: *mut c_void) -> c_int;
}

/// Same role as `RleBWT` (src/rle_bwt.rs:14-24); the index lives in HBM.
pub struct GpuRleBWT { raw: *mut MsbwtRle }

// Query entry points take &self and serialise inside the library; loads need &mut self.
unsafe impl Send for GpuRleBWT {}
unsafe impl Sync for GpuRleBWT {}

impl GpuRleBWT {
    pub fn new() -> Self { Self::with_bin_power(8) }

    pub fn with_bin_power(bin_power: u8) -> Self {
        let raw = unsafe { msbwt_rle_new(bin_power) };
        assert!(!raw.is_null(), "msbwt_rle_new: out of memory");
        GpuRleBWT { raw }
    }

    pub fn on_device(bin_power: u8, device: i32) -> Self {
        let raw = unsafe { msbwt_rle_new_on_device(bin_power, device) };
        assert!(!raw.is_null(), "msbwt_rle_new_on_device: out of memory");
        GpuRleBWT { raw }
    }

    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(msbwt_rle_last_error(self.raw)) }.to_string_lossy().into_owned()
    }

    /// Batch form of `count_kmer`: `kmers` holds n k-mers of k symbol codes each, row-major.
    pub fn count_kmers(&self, kmers: &[u8], k: usize) -> Vec<u64> {
        assert!(k == 0 || kmers.len() % k == 0);
        let n = if k == 0 { 0 } else { kmers.len() / k };
        let mut out = vec![0u64; n];
        let rc = unsafe { msbwt_rle_count_kmers(self.raw, kmers.as_ptr(), k, n, out.as_mut_ptr()) };
        if rc != MSBWT_OK { panic!("count_kmers: {}", self.last_error()); } // reference: assert!/panic
        out
    }

    /// Batch form of `constrain_range`.
    pub fn constrain_ranges(&self, syms: &[u8], ranges: &[BWTRange]) -> Vec<BWTRange> {
        assert_eq!(syms.len(), ranges.len());
        let l: Vec<u64> = ranges.iter().map(|r| r.l).collect();
        let h: Vec<u64> = ranges.iter().map(|r| r.h).collect();
        let (mut ol, mut oh) = (vec![0u64; l.len()], vec![0u64; l.len()]);
        let rc = unsafe {
            msbwt_rle_constrain_ranges(self.raw, syms.as_ptr(), l.as_ptr(), h.as_ptr(), l.len(),
                                       ol.as_mut_ptr(), oh.as_mut_ptr())
        };
        if rc != MSBWT_OK { panic!("constrain_ranges: {}", self.last_error()); }
        ol.into_iter().zip(oh).map(|(l, h)| BWTRange { l, h }).collect()
    }

    /// Counts of every k-mer window of `n` equal-length ASCII reads, forward strand and reverse
    /// complement, prepared on the GPU (no n x k query matrix).
    pub fn count_read_kmers(&self, reads: &[u8], read_len: usize, k: usize) -> (Vec<u64>, Vec<u64>) {
        assert!(read_len > 0 && reads.len() % read_len == 0 && k >= 1 && k <= read_len.min(64)); // the library takes 1 <= k <= 64
        let n = reads.len() / read_len;
        let w = read_len - k + 1;
        let (mut fwd, mut rc) = (vec![0u64; n * w], vec![0u64; n * w]);
        let code = unsafe {
            msbwt_rle_count_read_kmers(self.raw, reads.as_ptr(), read_len, n, k, 1, fwd.as_mut_ptr(), rc.as_mut_ptr())
        };
        if code != MSBWT_OK { panic!("count_read_kmers: {}", self.last_error()); }
        (fwd, rc)
    }

    pub fn set_table_depth(&mut self, depth: i32) {
        let rc = unsafe { msbwt_rle_set_table_depth(self.raw, depth) };
        if rc != MSBWT_OK { panic!("set_table_depth: {}", self.last_error()); }
    }

    /// `count_kmers` for k-mers over ACGT handed over as 2-bit words (`pack_2bit`): ceil(k / 32) words per k-mer.
    pub fn count_kmers_packed(&self, words: &[u64], k: usize) -> Vec<u64> {
        let per = if k > 32 { 2 } else { 1 };
        assert!(k >= 1 && k <= 64 && words.len() % per == 0);
        let n = words.len() / per;
        let mut out = vec![0u64; n];
        let rc = unsafe { msbwt_rle_count_kmers_packed(self.raw, words.as_ptr(), k, n, out.as_mut_ptr() as *mut c_void, 64) };
        if rc != MSBWT_OK { panic!("count_kmers_packed: {}", self.last_error()); }
        out
    }

    /// The same with 32-bit counts (half the bytes on the way back); panics if a count does not fit.
    pub fn count_kmers_packed_u32(&self, words: &[u64], k: usize) -> Vec<u32> {
        let per = if k > 32 { 2 } else { 1 };
        assert!(k >= 1 && k <= 64 && words.len() % per == 0);
        let n = words.len() / per;
        let mut out = vec![0u32; n];
        let rc = unsafe { msbwt_rle_count_kmers_packed(self.raw, words.as_ptr(), k, n, out.as_mut_ptr() as *mut c_void, 32) };
        if rc != MSBWT_OK { panic!("count_kmers_packed: {}", self.last_error()); }
        out
    }

    /// 1 = the library orders a batch itself whenever the pass applies; 0 and -1 (the default) = never (include/msbwt_hip.h).
    pub fn set_batch_order(&mut self, mode: i32) {
        let rc = unsafe { msbwt_rle_set_batch_order(self.raw, mode) };
        if rc != MSBWT_OK { panic!("set_batch_order: {}", self.last_error()); }
    }

    /// HBM the loaded index may hold (0 = no budget): plane blocks always, then pair blocks, then the deepest table that fits.
    pub fn set_memory_budget(&mut self, bytes: u64) {
        let rc = unsafe { msbwt_rle_set_memory_budget(self.raw, bytes) };
        if rc != MSBWT_OK { panic!("set_memory_budget: {}", self.last_error()); }
    }

    /// Sparse suffix table: -1 = automatic (default), 0 = off, 16..=31 = exactly that depth.  Results never change.
    pub fn set_sparse_table(&mut self, depth: i32) {
        let rc = unsafe { msbwt_rle_set_sparse_table(self.raw, depth) };
        if rc != MSBWT_OK { panic!("set_sparse_table: {}", self.last_error()); }
    }

    /// The k this index will mostly be asked about (0 = unknown): the automatic sparse table follows it (a table of d-mers serves k >= d).
    pub fn set_query_length(&mut self, k: i32) {
        let rc = unsafe { msbwt_rle_set_query_length(self.raw, k) };
        if rc != MSBWT_OK { panic!("set_query_length: {}", self.last_error()); }
    }

    pub fn query_length(&self) -> i32 { unsafe { msbwt_rle_get_query_length(self.raw) } }

    /// Two-tier form of the sparse table (entries for the suffixes that occur at least twice, filter bits for the rest -- read sets with
    /// errors): -1 = where the complete table of a depth does not fit (default), 0 = never, 1 = always.  Results never change.
    pub fn set_sparse_tiers(&mut self, mode: i32) {
        let rc = unsafe { msbwt_rle_set_sparse_tiers(self.raw, mode) };
        if rc != MSBWT_OK { panic!("set_sparse_tiers: {}", self.last_error()); }
    }

    pub fn sparse_tiers(&self) -> bool { unsafe { msbwt_rle_get_sparse_tiers(self.raw) != 0 } }

    /// The second, shallower sparse level (17-symbol suffixes, for k undeclared): -1 = automatic (default), 0 = never.
    pub fn set_sparse_second(&mut self, mode: i32) {
        let rc = unsafe { msbwt_rle_set_sparse_second(self.raw, mode) };
        if rc != MSBWT_OK { panic!("set_sparse_second: {}", self.last_error()); }
    }

    /// Depth of the sparse suffix table in HBM (0 = none) and how many distinct suffixes of that length occur.
    pub fn sparse_table(&self) -> (i32, u64) {
        let mut info = [0u64; 120];
        let rc = unsafe { msbwt_rle_sparse_table_info(self.raw, info.as_mut_ptr()) };
        if rc != MSBWT_OK { panic!("sparse_table_info: {}", self.last_error()); }
        (unsafe { msbwt_rle_get_sparse_table(self.raw) }, info[1])
    }
}

/// n k-mers of k symbol codes (A C G T = 1 2 3 5) -> 2-bit words: the k-mer as a base-4 number, first symbol most significant.
pub fn pack_2bit(kmers: &[u8], k: usize) -> Vec<u64> {
    assert!(k >= 1 && k <= 64 && kmers.len() % k == 0);
    let n = kmers.len() / k;
    let mut words = vec![0u64; n * if k > 32 { 2 } else { 1 }];
    let rc = unsafe { msbwt_kmers_pack_2bit(kmers.as_ptr(), k, n, words.as_mut_ptr()) };
    assert!(rc == MSBWT_OK, "msbwt_kmers_pack_2bit: a symbol outside A C G T (code {})", rc);
    words
}

/// One index replicated on several GPUs of a node; batches are sharded over the replicas inside
/// the library (one host thread + pinned pipeline per GPU), counts land in one Vec.
pub struct GpuRleBWTSet { replicas: Vec<GpuRleBWT> }

impl GpuRleBWTSet {
    /// `first` is loaded already; every further device gets a GPU -> GPU copy of its index
    /// (blocks, suffix table, pair index) over xGMI -- no second upload, no rebuild.
    pub fn replicate(first: GpuRleBWT, other_devices: &[i32]) -> Self {
        let mut replicas = vec![first];
        for &d in other_devices {
            let raw = unsafe { msbwt_rle_replicate(replicas[0].raw, d) };
            assert!(!raw.is_null(), "replicate: {}", replicas[0].last_error());
            replicas.push(GpuRleBWT { raw });
        }
        GpuRleBWTSet { replicas }
    }

    pub fn count_kmers(&self, kmers: &[u8], k: usize) -> Vec<u64> {
        assert!(k == 0 || kmers.len() % k == 0);
        let n = if k == 0 { 0 } else { kmers.len() / k };
        let raws: Vec<*const MsbwtRle> = self.replicas.iter().map(|r| r.raw as *const MsbwtRle).collect();
        let mut out = vec![0u64; n];
        let rc = unsafe { msbwt_rle_count_kmers_multi(raws.as_ptr(), raws.len(), kmers.as_ptr(), k, n, out.as_mut_ptr()) };
        if rc != MSBWT_OK { panic!("count_kmers_multi: code {}", rc); }
        out
    }
}

/// The key to sort a batch by (ascending) before it is counted: by the last 17 symbols of a k-mer, then leftwards -- the
/// order in which the suffix table and the BWT lay the queries' ranges out.  Counts never depend on it, speed does.
pub fn kmer_order_keys(kmers: &[u8], k: usize) -> Vec<u64> {
    assert!(k >= 1 && kmers.len() % k == 0);
    let n = kmers.len() / k;
    let mut keys = vec![0u64; n];
    let rc = unsafe { msbwt_kmer_order_keys(kmers.as_ptr(), k, n, keys.as_mut_ptr()) };
    assert!(rc == MSBWT_OK, "msbwt_kmer_order_keys: code {}", rc);
    keys
}

/// One rank of a one-process-per-GPU job: an RCCL communicator over the node's GPUs.  Rank 0 calls
/// `RankComm::unique_id()`, ships the 128 bytes to the other ranks (MPI, a file, the environment), every rank calls
/// `RankComm::init` with its GPU current; `GpuRleBWT::allgather_counts` is then the job's one exchange step.
pub struct RankComm { raw: *mut c_void, pub nranks: usize }

impl RankComm {
    pub fn unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        let rc = unsafe { msbwt_comm_get_unique_id(id.as_mut_ptr() as *mut c_void) };
        assert!(rc == MSBWT_OK, "msbwt_comm_get_unique_id: code {}", rc);
        id
    }
    pub fn init(nranks: usize, id: &[u8; 128], rank: usize) -> Self {
        let mut raw: *mut c_void = std::ptr::null_mut();
        let rc = unsafe { msbwt_comm_init_rank(&mut raw, nranks as c_int, id.as_ptr() as *const c_void, rank as c_int) };
        assert!(rc == MSBWT_OK && !raw.is_null(), "msbwt_comm_init_rank: code {}", rc);
        RankComm { raw, nranks }
    }
}

impl Drop for RankComm {
    fn drop(&mut self) { unsafe { msbwt_comm_destroy(self.raw); } }
}

impl GpuRleBWT {
    /// Every rank ends up with all ranks' counts: `d_all[r * n_mine + i]` = rank r's `d_mine[i]` (device pointers to
    /// u64; every rank passes the same `n_mine`).  `wire_bits` 16 or 32 narrows the counts on the wire; a count that
    /// does not fit is reported by the next `device_status` (repeat with 64).
    pub unsafe fn allgather_counts(&self, comm: &RankComm, d_mine: *const c_void, n_mine: usize, d_all: *mut c_void,
                                   wire_bits: i32, hip_stream: *mut c_void) {
        let rc = msbwt_rle_allgather_counts(self.raw, comm.raw, d_mine, n_mine, d_all, wire_bits as c_int, hip_stream);
        if rc != MSBWT_OK { panic!("allgather_counts: {}", self.last_error()); }
    }

    /// ONE batch counted and gathered as a pipeline: this rank's shard (`n_mine` k-mers of `k` symbol codes at `d_kmers`) is searched
    /// in `pieces` pieces while the counts of the finished pieces travel over RCCL on a second stream; `d_all[r * n_mine + i]` holds
    /// rank r's count i as `out_bits`-wide integers (64, or `wire_bits`: left as they arrived).
    pub unsafe fn count_kmers_allgather(&self, comm: &RankComm, d_kmers: *const c_void, k: usize, n_mine: usize, d_mine_counts: *mut c_void,
                                        d_all: *mut c_void, wire_bits: i32, out_bits: i32, pieces: i32, hip_stream: *mut c_void) {
        let rc = msbwt_rle_count_kmers_allgather_device(self.raw, comm.raw, d_kmers, k, n_mine, d_mine_counts, d_all, wire_bits as c_int,
                                                        out_bits as c_int, pieces as c_int, hip_stream);
        if rc != MSBWT_OK { panic!("count_kmers_allgather: {}", self.last_error()); }
    }
}

impl Default for GpuRleBWT { fn default() -> Self { Self::new() } }

impl Drop for GpuRleBWT {
    fn drop(&mut self) { unsafe { msbwt_rle_free(self.raw) } }
}

impl BWT for GpuRleBWT {
    fn load_vector(&mut self, bwt: Vec<u8>) {
        let rc = unsafe { msbwt_rle_load_vector(self.raw, bwt.as_ptr(), bwt.len()) };
        // the trait method is infallible; the reference panics on a symbol code >= 6
        if rc != MSBWT_OK { panic!("load_vector: {}", self.last_error()); }
    }

    fn load_numpy_file(&mut self, filename: &str) -> io::Result<()> {
        let path = CString::new(filename).map_err(|e| io::Error::new(io::ErrorKind::InvalidInput, e))?;
        match unsafe { msbwt_rle_load_numpy_file(self.raw, path.as_ptr()) } {
            MSBWT_OK => Ok(()),
            MSBWT_ERR_IO => Err(io::Error::new(io::ErrorKind::Other, self.last_error())),
            MSBWT_ERR_UNEXPECTED_EOF => Err(io::Error::new(io::ErrorKind::UnexpectedEof, self.last_error())),
            _ => panic!("{}", self.last_error()), // malformed header: the reference panics too
        }
    }

    #[inline]
    fn get_symbol_count(&self, symbol: u8) -> u64 {
        assert!((symbol as usize) < VC_LEN); // reference: array index panic
        unsafe { msbwt_rle_get_symbol_count(self.raw, symbol) }
    }

    #[inline]
    fn get_total_size(&self) -> u64 { unsafe { msbwt_rle_get_total_size(self.raw) } }

    unsafe fn constrain_range(&self, sym: u8, input_range: &BWTRange) -> BWTRange {
        let (mut l, mut h) = (0u64, 0u64);
        let rc = msbwt_rle_constrain_range(self.raw, sym, input_range.l, input_range.h, &mut l, &mut h);
        if rc != MSBWT_OK { panic!("constrain_range: {}", self.last_error()); }
        BWTRange { l, h }
    }

    // overrides the default body (src/msbwt_core.rs:124-161): the whole k-step loop runs in one launch
    fn count_kmer(&self, kmer: &[u8]) -> u64 {
        let mut out = 0u64;
        let rc = unsafe { msbwt_rle_count_kmer(self.raw, kmer.as_ptr(), kmer.len(), &mut out) };
        if rc != MSBWT_OK { panic!("count_kmer: {}", self.last_error()); }
        out
    }
}
