//! Drop-in replacements for the `pa_bitpacking` operators the A*PA2 block engine calls
//! (`astarpa2/src/blocks.rs:112,631,719-724`), computed on an MI355X by `libastarpa_c_hip.so`
//! (`include/pa_bitpacking_hip.h`).
//!
//! NOT COMPILED IN THIS REPOSITORY (no Rust toolchain in the build image).  The C side of every declaration is pinned by
//! `tests/c_abi/layout_check.c` and `tests/test_capi_symbols.py`.
//!
//! Rust tuples and tuple structs are not `repr(C)`.  `V(u64, u64)`, `Bits(u64, u64)` and `(u64, u64)` are two consecutive
//! `u64` in practice; the `const` assertions below check size and alignment, and the recommended two-line change to the
//! reference is `#[repr(C)]` on `V` (`pa-bitpacking/src/encoding.rs:5`) and `Bits` (`profile.rs:92`).
use pa_bitpacking::{Bits, V};
use pa_types::Cost;
use std::os::raw::c_char;

#[link(name = "astarpa_c_hip")]
extern "C" {
    /// `BitProfile::build` (profile.rs:112-133). `a2` has `2 n` u64, `b2` has `2 ceil(m / 64)` u64.
    pub fn pa_bp_profile_build(a: *const u8, n: usize, b: *const u8, m: usize, a2: *mut u64, b2: *mut u64) -> i32;
    /// `simd::compute` (simd.rs:98-226): returns the sum of the bottom deltas, `i32::MIN` on error.
    pub fn pa_bp_compute(a2: *const u64, n: usize, b2: *const u64, w: usize, h2: *mut u64, v2: *mut u64, exact_end: i32) -> i32;
    /// `simd::fill` (simd.rs:326-437): additionally `values[(i * w + j) * 2 + {0, 1}]`.
    pub fn pa_bp_fill(a2: *const u64, n: usize, b2: *const u64, w: usize, h2: *mut u64, v2: *mut u64, values: *mut u64) -> i32;
    /// Device-resident handles: the sequences, the profile and the row of horizontal deltas (`Blocks::h`) stay on the GPU between
    /// the calls of one pair; `h_mode` = `HMode::{None, Input, Update, Output}` as 0..3 (blocks.rs:665-671).
    pub fn pa_bp_ctx_create(a: *const u8, n: usize, b: *const u8, m: usize) -> *mut core::ffi::c_void;
    pub fn pa_bp_ctx_compute(ctx: *mut core::ffi::c_void, i0: i32, i1: i32, w0: usize, w1: usize, v: *mut u64, h_mode: i32,
                             sum_out: *mut i32) -> i32;
    pub fn pa_bp_ctx_fill(ctx: *mut core::ffi::c_void, i0: i32, i1: i32, w0: usize, w1: usize, v: *mut u64, values: *mut u64,
                          h_bottom: *mut i8) -> i32;
    pub fn pa_bp_ctx_destroy(ctx: *mut core::ffi::c_void);
    pub fn pa_last_error() -> *const c_char;
    pub fn pa_device_count() -> i32;
    pub fn pa_set_device(device: i32) -> i32;
}

const _: () = assert!(std::mem::size_of::<(u64, u64)>() == 16 && std::mem::align_of::<(u64, u64)>() == 8);
const _: () = assert!(std::mem::size_of::<V>() == 16 && std::mem::size_of::<Bits>() == 16);

fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(pa_last_error()) }.to_string_lossy().into_owned()
}

/// `pa_bitpacking::BitProfile::build(a, b)` (profile.rs:112).
pub fn build(a: &[u8], b: &[u8]) -> (Vec<Bits>, Vec<Bits>) {
    let w = (b.len() + 63) / 64;
    let mut pa = vec![Bits(0, 0); a.len()];
    let mut pb = vec![Bits(0, 0); w];
    let rc = unsafe { pa_bp_profile_build(a.as_ptr(), a.len(), b.as_ptr(), b.len(), pa.as_mut_ptr() as *mut u64, pb.as_mut_ptr() as *mut u64) };
    assert!(rc == 0, "pa_bp_profile_build: {}", last_error()); // the reference panics on a non-ACGT base as well
    (pa, pb)
}

/// `pa_bitpacking::simd::compute::<N, (u64, u64), L>` (simd.rs:98-104).
pub fn compute(a: &[Bits], b: &[Bits], h: &mut [(u64, u64)], v: &mut [V], exact_end: bool) -> Cost {
    assert_eq!(a.len(), h.len());
    assert_eq!(b.len(), v.len());
    let r = unsafe {
        pa_bp_compute(a.as_ptr() as *const u64, a.len(), b.as_ptr() as *const u64, b.len(), h.as_mut_ptr() as *mut u64,
                      v.as_mut_ptr() as *mut u64, exact_end as i32)
    };
    assert!(r != i32::MIN, "pa_bp_compute: {}", last_error());
    r
}

/// `pa_bitpacking::simd::fill::<N, (u64, u64), L>` (simd.rs:326-333): `values[i][j]` from the flat `n x w` buffer.
pub fn fill(a: &[Bits], b: &[Bits], h: &mut [(u64, u64)], v: &mut [V], _exact_end: bool, values: &mut [Vec<V>]) -> Cost {
    assert_eq!(a.len(), h.len());
    assert_eq!(b.len(), v.len());
    assert_eq!(values.len(), h.len());
    let (n, w) = (a.len(), b.len());
    let mut flat = vec![V::zero(); n * w];
    let r = unsafe {
        pa_bp_fill(a.as_ptr() as *const u64, n, b.as_ptr() as *const u64, w, h.as_mut_ptr() as *mut u64, v.as_mut_ptr() as *mut u64,
                   flat.as_mut_ptr() as *mut u64)
    };
    assert!(r != i32::MIN, "pa_bp_fill: {}", last_error());
    for (i, vv) in values.iter_mut().enumerate() {
        vv.clear();
        vv.extend_from_slice(&flat[i * w..(i + 1) * w]);
    }
    r
}

/// `HMode` of the block engine (`astarpa2/src/blocks.rs:665-671`), as the C ABI numbers it.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
#[repr(i32)]
pub enum HMode {
    None = 0,
    Input = 1,
    Update = 2,
    Output = 3,
}

/// What `Blocks` keeps next to its `Vec<Block>` when the kernels run on the GPU: one device-resident context per pair.  The
/// sequences, their profile and the persistent row of horizontal deltas (`Blocks::h`, blocks.rs:103-105) live in HBM between the
/// block calls of one alignment; a call moves only the `v` words of its rectangle.
///
/// In the reference, `Blocks::new` (blocks.rs:110-128) builds the `BitProfile` and allocates `h`; with this shim it creates a
/// `HipBlocksCtx` instead, and the free function `compute_block` (blocks.rs:686-748) / the `fill` call of `fill_with_blocks`
/// (blocks.rs:627-648) become the two methods below -- same arguments (`i_range`, `v_range`, `v`, `HMode`), same return value.
pub struct HipBlocksCtx {
    raw: *mut core::ffi::c_void,
}

impl HipBlocksCtx {
    /// `BitProfile::build(a, b)` + the allocation of `h` (blocks.rs:112-123), on the device.
    pub fn new(a: &[u8], b: &[u8]) -> Self {
        let raw = unsafe { pa_bp_ctx_create(a.as_ptr(), a.len(), b.as_ptr(), b.len()) };
        assert!(!raw.is_null(), "pa_bp_ctx_create: {}", last_error()); // the reference panics on a non-ACGT base as well
        HipBlocksCtx { raw }
    }

    /// `compute_block(params, a, b, h, i_range, v_range, v, mode, ..)` (blocks.rs:686-748): columns `(i_range.0, i_range.1]` of the
    /// matrix (characters `a[i_range.0 .. i_range.1)`), 64-row words `v_range` of `b`; `v` holds exactly those words and is updated
    /// in place; returns the sum of the bottom-row deltas.  `HMode::Input / Update` read the stored row, `Update / Output` store the
    /// bottom row back (blocks.rs:728-747).
    pub fn compute_block(&mut self, i_range: (i32, i32), v_range: std::ops::Range<usize>, v: &mut [V], mode: HMode) -> Cost {
        assert_eq!(v.len(), v_range.len());
        let mut sum: i32 = 0;
        let rc = unsafe {
            pa_bp_ctx_compute(self.raw, i_range.0, i_range.1, v_range.start, v_range.end, v.as_mut_ptr() as *mut u64, mode as i32, &mut sum)
        };
        assert!(rc == 0, "pa_bp_ctx_compute: {}", last_error());
        sum
    }

    /// The `pa_bitpacking::simd::fill` call of `fill_with_blocks` (blocks.rs:627-648): top row `+1`, `values[i][j]` = `V` of word
    /// `v_range.start + j` after column `i_range.0 + i + 1`; `h_bottom[i]` (if given) = the bottom-row delta of that column.  The
    /// stored row is not touched.
    pub fn fill_block(&mut self, i_range: (i32, i32), v_range: std::ops::Range<usize>, v: &mut [V], values: &mut [Vec<V>],
                      h_bottom: Option<&mut [i8]>) {
        let (n, w) = ((i_range.1 - i_range.0) as usize, v_range.len());
        assert_eq!(v.len(), w);
        assert_eq!(values.len(), n);
        let mut flat = vec![V::zero(); n * w];
        let hb = match h_bottom {
            Some(h) => {
                assert_eq!(h.len(), n);
                h.as_mut_ptr()
            }
            None => std::ptr::null_mut(),
        };
        let rc = unsafe {
            pa_bp_ctx_fill(self.raw, i_range.0, i_range.1, v_range.start, v_range.end, v.as_mut_ptr() as *mut u64, flat.as_mut_ptr() as *mut u64, hb)
        };
        assert!(rc == 0, "pa_bp_ctx_fill: {}", last_error());
        for (i, vv) in values.iter_mut().enumerate() {
            vv.clear();
            vv.extend_from_slice(&flat[i * w..(i + 1) * w]);
        }
    }
}

impl Drop for HipBlocksCtx {
    fn drop(&mut self) {
        unsafe { pa_bp_ctx_destroy(self.raw) }
    }
}

// A context belongs to the thread (and the device) that created it (`pa_set_device` is per thread).
// (no `Send` / `Sync`: raw pointer)
