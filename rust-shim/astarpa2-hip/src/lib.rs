//! `pa_types::Aligner` backed by `pa_align` of `libastarpa_c_hip.so` (`include/pa_astarpa2.h`): the whole A*PA2 aligner on an
//! MI355X (band doubling, sparse blocks, DT-trace, SH / GCSH, incremental doubling; `astarpa2/src/lib.rs:122-214`).
//!
//! NOT COMPILED IN THIS REPOSITORY (no Rust toolchain in the build image).  The structs below mirror `include/pa_astarpa2.h`
//! field for field, in order; `tests/c_abi/layout_check.c` pins the offsets on the C side and
//! `tests/test_capi_symbols.py::test_rust_mirror_matches_header` compares the field lists of this file with the header.
use pa_types::{Aligner, Cigar, CigarElem, CigarOp, Cost, Seq};
use std::os::raw::c_char;

/// `pa_block_params` = `BlockParams` (astarpa2/src/blocks.rs:31-60).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PaBlockParams {
    pub sparse: i32,
    pub simd: i32,
    pub no_ilp: i32,
    pub incremental_doubling: i32,
    pub dt_trace: i32,
    pub max_g: i32,
    pub fr_drop: i32,
}

/// `pa_astarpa2_params` = `AstarPa2Params` (astarpa2/src/params.rs:8-42).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PaAstarPa2Params {
    pub domain: i32,
    pub heuristic: i32,
    pub heuristic_k: i32,
    pub heuristic_p: i32,
    pub doubling: i32,
    pub doubling_start: i32,
    pub factor: f32,
    pub delta: f32,
    pub block_width: i32,
    pub front: PaBlockParams,
    pub sparse_h: i32,
    pub prune: i32,
}

/// `pa_astarpa2_stats` = `AstarPa2Stats` + `BlockStats` + `TraceStats` (domain.rs:31-43, blocks.rs:76-84, trace.rs:3-14).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PaAstarPa2Stats {
    pub num_blocks: u64,
    pub num_incremental_blocks: u64,
    pub computed_lanes: u64,
    pub unique_lanes: u64,
    pub dt_trace_tries: u64,
    pub dt_trace_success: u64,
    pub dt_trace_fallback: u64,
    pub fill_tries: u64,
    pub fill_success: u64,
    pub fill_fallback: u64,
    pub f_max_tries: u64,
    pub sanity_violations: u64,
    pub t_compute: f64,
    pub t_dt: f64,
    pub t_fill: f64,
    pub t_precomp: f64,
    pub t_j_range: f64,
    pub t_fixed_j_range: f64,
    pub t_pruning: f64,
    pub t_contours_update: f64,
}

#[link(name = "astarpa_c_hip")]
extern "C" {
    pub fn pa_params_nw(p: *mut PaAstarPa2Params);
    pub fn pa_params_simple(p: *mut PaAstarPa2Params);
    pub fn pa_params_full(p: *mut PaAstarPa2Params);
    pub fn pa_align(a: *const u8, a_len: usize, b: *const u8, b_len: usize, params: *const PaAstarPa2Params, trace: i32,
                    cost_out: *mut i32, cigar_out: *mut *mut c_char, stats_out: *mut PaAstarPa2Stats) -> i32;
    pub fn astarpa_free_cigar(cigar: *mut u8);
    pub fn pa_last_error() -> *const c_char;
    // many pairs at once (include/pa_bitpacking_hip.h): cost + CIGAR of every pair from one call, with the traceback options of
    // `trace_params.front` (NULL: re-fill only), and the same sharded over several GPUs by host threads inside the library
    pub fn pa_batch_create_trace_params(a: *const *const u8, a_len: *const usize, b: *const *const u8, b_len: *const usize, pairs: usize,
                                        trace_params: *const PaAstarPa2Params) -> *mut core::ffi::c_void;
    pub fn pa_batch_align(plan: *mut core::ffi::c_void, cost_out: *mut i32, cigar_out: *mut *mut c_char, forward_ms: *mut f32,
                          trace_ms: *mut f32) -> i32;
    // ... without one malloc'ed string per pair: text_out[i] .. + text_len_out[i] (no NUL) inside the plan, valid until the next call / destroy
    pub fn pa_batch_align_view(plan: *mut core::ffi::c_void, cost_out: *mut i32, text_out: *mut *const c_char, text_len_out: *mut u32,
                               forward_ms: *mut f32, trace_ms: *mut f32) -> i32;
    pub fn pa_batch_destroy(plan: *mut core::ffi::c_void);
    pub fn pa_batch_align_multi(a: *const *const u8, a_len: *const usize, b: *const *const u8, b_len: *const usize, pairs: usize,
                                devices: *const i32, ndevices: i32, cost_out: *mut i32, cigar_out: *mut *mut c_char) -> i32;
    // batched A*PA2 (the `simple` preset and its relatives): one wavefront runs a pair's whole band search; cost, CIGAR and
    // statistics are what a loop over `align_with_stats` returns
    pub fn pa_batch_create_params(a: *const *const u8, a_len: *const usize, b: *const *const u8, b_len: *const usize, pairs: usize,
                                  params: *const PaAstarPa2Params) -> *mut core::ffi::c_void;
    pub fn pa_batch_pair_stats(plan: *const core::ffi::c_void, stats_out: *mut PaAstarPa2Stats) -> i32;
    pub fn pa_batch_params_supported(params: *const PaAstarPa2Params) -> i32;
    /// Diagnostics (round 5): how the half-wave blocks of the last forward pass met -- out4 = fused, served by a partner, alone, withdrawn.
    pub fn pa_batch_rdv_stats(plan: *const core::ffi::c_void, out4: *mut u64) -> i32;
    /// Pairs whose band left their window of the block-column store and were aligned again / the largest store such a second round held.
    pub fn pa_batch_window_retries(plan: *const core::ffi::c_void) -> usize;
    pub fn pa_batch_window_retry_bytes(plan: *const core::ffi::c_void) -> f64;
    pub fn pa_batch_align_multi_params(a: *const *const u8, a_len: *const usize, b: *const *const u8, b_len: *const usize, pairs: usize,
                                       devices: *const i32, ndevices: i32, params: *const PaAstarPa2Params, cost_out: *mut i32,
                                       cigar_out: *mut *mut c_char, stats_out: *mut PaAstarPa2Stats) -> i32;
    pub fn pa_runtime_hints() -> i32;
    pub fn pa_release_pools();
    /// Round 5: callers that are inside `pa_align` (or an astarpa-c symbol) at the same time are combined into one batch on the GPU;
    /// calls served that way so far and the batches they went out in (include/pa_astarpa2.h).
    pub fn pa_combine_stats(calls: *mut u64, batches: *mut u64);
}

/// The text form of `Cigar::to_string` (count omitted when 1; `=`, `X`, `I`, `D`; astarpa-c/example.cpp:16 `"=I4=X="`).
pub fn parse_cigar(s: &str) -> Cigar {
    let mut ops = Vec::new();
    let mut cnt: i32 = 0;
    let mut have = false;
    for ch in s.bytes() {
        if ch.is_ascii_digit() {
            cnt = cnt * 10 + (ch - b'0') as i32;
            have = true;
            continue;
        }
        let op = match ch {
            b'=' => CigarOp::Match,
            b'X' => CigarOp::Sub,
            b'I' => CigarOp::Ins,
            b'D' => CigarOp::Del,
            _ => panic!("unexpected CIGAR character {}", ch as char),
        };
        ops.push(CigarElem { op, cnt: if have { cnt } else { 1 } });
        cnt = 0;
        have = false;
    }
    Cigar { ops }
}

#[derive(Debug, Clone, Copy)]
pub struct HipAstarPa2 {
    pub params: PaAstarPa2Params,
    pub trace: bool,
}

impl HipAstarPa2 {
    /// `AstarPa2Params::simple().make_aligner(trace)` (params.rs:70-96,132).
    pub fn simple(trace: bool) -> Self {
        let mut params = PaAstarPa2Params::default();
        unsafe { pa_params_simple(&mut params) };
        Self { params, trace }
    }
    /// `AstarPa2Params::full().make_aligner(trace)` (params.rs:98-128).
    pub fn full(trace: bool) -> Self {
        let mut params = PaAstarPa2Params::default();
        unsafe { pa_params_full(&mut params) };
        Self { params, trace }
    }
    /// `AstarPa2Params::nw().make_aligner(trace)` (params.rs:46-68).
    pub fn nw(trace: bool) -> Self {
        let mut params = PaAstarPa2Params::default();
        unsafe { pa_params_nw(&mut params) };
        Self { params, trace }
    }

    /// `AstarPa2StatsAligner::align_with_stats` (astarpa2/src/lib.rs:200-208).
    pub fn align_with_stats(&mut self, a: Seq, b: Seq) -> (Cost, Option<Cigar>, PaAstarPa2Stats) {
        let mut cost = 0i32;
        let mut ptr: *mut c_char = std::ptr::null_mut();
        let mut stats = PaAstarPa2Stats::default();
        let rc = unsafe { pa_align(a.as_ptr(), a.len(), b.as_ptr(), b.len(), &self.params, self.trace as i32, &mut cost, &mut ptr, &mut stats) };
        if rc != 0 {
            // the reference panics on invalid input / internal inconsistencies; keep that contract
            panic!("pa_align failed ({}): {}", rc, unsafe { std::ffi::CStr::from_ptr(pa_last_error()) }.to_string_lossy());
        }
        let cigar = (!ptr.is_null()).then(|| {
            let s = unsafe { std::ffi::CStr::from_ptr(ptr) }.to_str().unwrap().to_owned();
            unsafe { astarpa_free_cigar(ptr as *mut u8) };
            parse_cigar(&s)
        });
        (cost, cigar, stats)
    }
}

impl HipAstarPa2 {
    /// The loop of pa-bin (`for (a, b) in pairs { aligner.align(a, b) }`, pa-bin/src/main.rs:24-35) as ONE call: every pair's band
    /// search runs on the GPU side by side (`pa_batch_create_params` + `pa_batch_align_view`); parameters outside the batched family
    /// (and `trace == false`) fall back to the loop.
    pub fn align_many(&mut self, pairs: &[(Seq, Seq)]) -> Vec<(Cost, Option<Cigar>, PaAstarPa2Stats)> {
        let n = pairs.len();
        let ap: Vec<*const u8> = pairs.iter().map(|p| p.0.as_ptr()).collect();
        let bp: Vec<*const u8> = pairs.iter().map(|p| p.1.as_ptr()).collect();
        let al: Vec<usize> = pairs.iter().map(|p| p.0.len()).collect();
        let bl: Vec<usize> = pairs.iter().map(|p| p.1.len()).collect();
        let plan = if self.trace { unsafe { pa_batch_create_params(ap.as_ptr(), al.as_ptr(), bp.as_ptr(), bl.as_ptr(), n, &self.params) } } else { std::ptr::null_mut() };
        if plan.is_null() {
            return pairs.iter().map(|(a, b)| self.align_with_stats(a, b)).collect();
        }
        let mut costs = vec![0i32; n];
        let mut ptrs: Vec<*const c_char> = vec![std::ptr::null(); n];
        let mut lens = vec![0u32; n];
        let mut stats = vec![PaAstarPa2Stats::default(); n];
        // the texts stay in the plan's host buffer (pa_batch_align_view): parsed into `Cigar`s before the plan goes
        let rc = unsafe { pa_batch_align_view(plan, costs.as_mut_ptr(), ptrs.as_mut_ptr(), lens.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) };
        let rc2 = if rc == 0 { unsafe { pa_batch_pair_stats(plan, stats.as_mut_ptr()) } } else { rc };
        if rc2 != 0 {
            unsafe { pa_batch_destroy(plan) };
            panic!("pa_batch_align_view failed ({}): {}", rc2, unsafe { std::ffi::CStr::from_ptr(pa_last_error()) }.to_string_lossy());
        }
        let out = (0..n)
            .map(|i| {
                let bytes: &[u8] = if ptrs[i].is_null() { &[] } else { unsafe { std::slice::from_raw_parts(ptrs[i] as *const u8, lens[i] as usize) } };
                (costs[i], Some(parse_cigar(std::str::from_utf8(bytes).unwrap())), stats[i])
            })
            .collect();
        unsafe { pa_batch_destroy(plan) };
        out
    }
}

impl Aligner for HipAstarPa2 {
    fn align(&mut self, a: Seq, b: Seq) -> (Cost, Option<Cigar>) {
        let (cost, cigar, _) = self.align_with_stats(a, b);
        (cost, cigar)
    }
}
