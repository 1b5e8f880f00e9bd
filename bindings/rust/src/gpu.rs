//! Safe layer over `ffi`: context, device buffers, frame batches, and the reference crate's traits.
//!
//! Error convention.  The reference returns `VoxBoxResult` from a few methods and panics (asserts, index out of
//! bounds, `partial_cmp().unwrap()` on NaN) everywhere else.  The library reports both per frame in a status
//! array ([`FrameStatus`]) so that one bad frame never aborts a batch.  The batched methods here hand that array
//! to the caller; the drop-in per-frame trait methods turn it back into the reference's behaviour for that frame:
//! `Err(VoxBoxError::LPC(..))` / `Err(VoxBoxError::Polynomial(..))` where the trait returns a result, a panic
//! where the reference panics.  API misuse and HIP runtime failures are [`GpuError`]; inside the trait methods
//! (whose signatures have no error channel) they panic, as the reference's own asserts do.

use std::cell::RefCell;
use std::ffi::CStr;
use std::fmt;
use std::marker::PhantomData;
use std::os::raw::c_void;
use std::ptr;

use num_complex::Complex;
use vox_box::error::{VoxBoxError, VoxBoxResult};
use vox_box::periodic::{Autocorrelate, LagType, Pitch, Pitched};
use vox_box::spectrum::{Resonance, ToResonance, LPC, MFCC};

use crate::ffi;

// ---------------------------------------------------------------------------------------------------------------
// errors and per-frame status
// ---------------------------------------------------------------------------------------------------------------

/// API misuse (`VBX_E_INVALID`), HIP runtime failure (`VBX_E_RUNTIME`) or no gfx950 device (`VBX_E_NODEVICE`).
#[derive(Debug, Clone)]
pub struct GpuError {
    pub code: i32,
    pub message: String,
}

impl fmt::Display for GpuError {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "voxbox_hip error {}: {}", self.code, self.message)
    }
}

impl std::error::Error for GpuError {}

pub type GpuResult<T> = Result<T, GpuError>;

/// Per-frame status codes of the library (`VBX_FRAME_*`): what the reference does on that frame.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum FrameStatus {
    /// the reference returns normally
    Ok,
    /// `Err(VoxBoxError::LPC("Denum was <= 0.0"))`, src/spectrum.rs:123-125
    Lpc,
    /// `Err(VoxBoxError::Polynomial(..))`, src/polynomial.rs:95,123
    Polynomial,
    /// `partial_cmp().unwrap()` on a NaN strength panics, src/periodic.rs:453
    NaN,
    /// any other panic of the reference (index out of bounds, assert)
    Panic,
}

impl FrameStatus {
    pub fn from_code(code: i32) -> FrameStatus {
        match code {
            0 => FrameStatus::Ok,
            1 => FrameStatus::Lpc,
            2 => FrameStatus::Polynomial,
            3 => FrameStatus::NaN,
            _ => FrameStatus::Panic,
        }
    }

    /// The reference's behaviour on this frame for a method that returns `VoxBoxResult`: `Ok`, the `Err` the
    /// reference returns, or the panic the reference raises.
    pub fn into_result(self) -> VoxBoxResult<()> {
        match self {
            FrameStatus::Ok => Ok(()),
            FrameStatus::Lpc => Err(VoxBoxError::LPC("Denum was <= 0.0")),
            FrameStatus::Polynomial => Err(VoxBoxError::Polynomial("Polynomial has degree < 1 or no roots found")),
            FrameStatus::NaN => panic!("vox_box: partial_cmp().unwrap() on a NaN value (src/periodic.rs:453)"),
            FrameStatus::Panic => panic!("vox_box: the reference panics on this frame (index out of bounds / assert)"),
        }
    }

    /// For methods without an error channel: every status but `Ok` is a panic, as in the reference.
    pub fn unwrap_or_panic(self) {
        if let Err(e) = self.into_result() {
            panic!("vox_box: {:?}", e);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// context and device memory
// ---------------------------------------------------------------------------------------------------------------

/// One library context: a HIP stream on one MI355X, cached tables, scratch.  Not `Sync`: one host thread at a time
/// (contexts are cheap; create one per thread).
pub struct Gpu {
    raw: *mut ffi::VbxCtx,
}

impl Gpu {
    /// `device`: HIP device ordinal.  Fails when there is no usable gfx950 device (no CPU fallback).
    pub fn new(device: i32) -> GpuResult<Gpu> {
        let mut raw: *mut ffi::VbxCtx = ptr::null_mut();
        let rc = unsafe { ffi::vbx_ctx_create(&mut raw, device, ptr::null_mut()) };
        if rc != ffi::VBX_SUCCESS {
            return Err(GpuError { code: rc, message: last_error(ptr::null()) });
        }
        Ok(Gpu { raw })
    }

    /// Launch on a stream the caller owns (`hipStream_t`).
    ///
    /// # Safety
    /// `hip_stream` must be a valid `hipStream_t` of `device` that outlives the context.
    pub unsafe fn with_stream(device: i32, hip_stream: *mut c_void) -> GpuResult<Gpu> {
        let mut raw: *mut ffi::VbxCtx = ptr::null_mut();
        let rc = ffi::vbx_ctx_create(&mut raw, device, hip_stream);
        if rc != ffi::VBX_SUCCESS {
            return Err(GpuError { code: rc, message: last_error(ptr::null()) });
        }
        Ok(Gpu { raw })
    }

    pub fn raw(&self) -> *mut ffi::VbxCtx {
        self.raw
    }

    pub fn check(&self, rc: i32) -> GpuResult<()> {
        if rc == ffi::VBX_SUCCESS {
            Ok(())
        } else {
            Err(GpuError { code: rc, message: last_error(self.raw) })
        }
    }

    /// Waits for everything queued on the context's stream.
    pub fn sync(&self) -> GpuResult<()> {
        self.check(unsafe { ffi::vbx_sync(self.raw) })
    }

    /// Uninitialised device memory for `len` elements.
    pub fn alloc<T: Copy>(&self, len: usize) -> GpuResult<DeviceBuf<T>> {
        let mut p: *mut c_void = ptr::null_mut();
        let bytes = len.max(1) * std::mem::size_of::<T>();
        self.check(unsafe { ffi::vbx_malloc(self.raw, &mut p, bytes) })?;
        Ok(DeviceBuf { gpu: self, ptr: p as *mut T, len, _own: PhantomData })
    }

    /// Host slice -> HBM.
    pub fn upload<T: Copy>(&self, host: &[T]) -> GpuResult<DeviceBuf<T>> {
        let buf = self.alloc::<T>(host.len())?;
        if !host.is_empty() {
            self.check(unsafe {
                ffi::vbx_memcpy_h2d(self.raw, buf.ptr as *mut c_void, host.as_ptr() as *const c_void, host.len() * std::mem::size_of::<T>())
            })?;
        }
        Ok(buf)
    }

    /// `sample::window` tables built on the host with the reference's recurrences (`VBX_WINDOW_*`).
    pub fn window_table(kind: i32, n: usize) -> GpuResult<Vec<f64>> {
        let mut t = vec![0f64; n];
        let rc = unsafe { ffi::vbx_window_table_f64(kind, n, t.as_mut_ptr()) };
        if rc != ffi::VBX_SUCCESS {
            return Err(GpuError { code: rc, message: last_error(ptr::null()) });
        }
        Ok(t)
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { ffi::vbx_ctx_destroy(self.raw) }
    }
}

fn last_error(ctx: *const ffi::VbxCtx) -> String {
    let p = unsafe { ffi::vbx_last_error(ctx) };
    if p.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

/// `len` elements of `T` in HBM, freed on drop.
pub struct DeviceBuf<'g, T: Copy> {
    gpu: &'g Gpu,
    ptr: *mut T,
    len: usize,
    _own: PhantomData<T>,
}

impl<'g, T: Copy> DeviceBuf<'g, T> {
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    pub fn as_ptr(&self) -> *const T {
        self.ptr as *const T
    }
    pub fn as_mut_ptr(&self) -> *mut T {
        self.ptr
    }
    /// HBM -> host (waits for the stream).
    pub fn to_vec(&self) -> GpuResult<Vec<T>> {
        let mut v: Vec<T> = Vec::with_capacity(self.len);
        if self.len > 0 {
            self.gpu.check(unsafe {
                ffi::vbx_memcpy_d2h(self.gpu.raw, v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * std::mem::size_of::<T>())
            })?;
        }
        unsafe { v.set_len(self.len) };
        Ok(v)
    }
}

impl<'g, T: Copy> Drop for DeviceBuf<'g, T> {
    fn drop(&mut self) {
        unsafe {
            ffi::vbx_free(self.gpu.raw, self.ptr as *mut c_void);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// frame batches
// ---------------------------------------------------------------------------------------------------------------

#[derive(Clone, Copy, PartialEq)]
struct PitchKey {
    sample_rate: u64,
    threshold: u64,
    min: u64,
    max: u64,
    kmax: usize,
}

struct PitchRows {
    key: PitchKey,
    cand: Vec<ffi::VbxPitch>, // [F, kmax]
    count: Vec<i32>,
    status: Vec<i32>,
}

#[derive(Default)]
struct Cache {
    autocorr: Option<(usize, Vec<f64>)>,                 // n_lags, [F, n_lags]
    lpc: Option<(usize, Vec<f64>, Vec<f64>)>,            // n_coeffs, ac [F, p+1], kc [F, p]
    burg: Option<(usize, Vec<f64>, Vec<i32>)>,           // n_coeffs, [F, p], status
    mfcc: Option<((usize, u64, u64, u64), Vec<f64>, Vec<i32>)>,
    pitch: Option<PitchRows>,
}

/// All frames of a recording in HBM: `window::Windower::{hanning, rectangle}(samples, bin, hop)` as a strided VIEW
/// of the contiguous samples (no frame copies), or a dense `[F, N]` array.  `window` is the Windower's multiplier
/// table, applied on load by every kernel -- the frames the reference's traits receive are already windowed.
pub struct FrameBatch<'g> {
    gpu: &'g Gpu,
    samples: DeviceBuf<'g, f64>,
    window: Option<DeviceBuf<'g, f64>>,
    n_frames: usize,
    frame_len: usize,
    stride: usize,
    pitch_kmax: usize,
    cache: RefCell<Cache>,
}

impl<'g> FrameBatch<'g> {
    fn new(gpu: &'g Gpu, samples: DeviceBuf<'g, f64>, window: Option<DeviceBuf<'g, f64>>, n_frames: usize, frame_len: usize, stride: usize) -> FrameBatch<'g> {
        FrameBatch {
            gpu,
            samples,
            window,
            n_frames,
            frame_len,
            stride,
            pitch_kmax: ffi::vbx_pitch_max_candidates(frame_len),
            cache: RefCell::new(Cache::default()),
        }
    }

    /// `window::Windower::hanning(samples, bin, hop)` (examples/pitch_detection.rs:23).
    pub fn windower_hanning(gpu: &'g Gpu, samples: &[f64], bin: usize, hop: usize) -> GpuResult<FrameBatch<'g>> {
        let table = Gpu::window_table(ffi::VBX_WINDOW_HANNING, bin)?;
        let n_frames = unsafe { ffi::vbx_frame_count(samples.len(), bin, hop) };
        Ok(FrameBatch::new(gpu, gpu.upload(samples)?, Some(gpu.upload(&table)?), n_frames, bin, hop))
    }

    /// `window::Windower::rectangle(samples, bin, hop)` (tests/lib.rs:71): what `find_formants` expects.
    pub fn windower_rectangle(gpu: &'g Gpu, samples: &[f64], bin: usize, hop: usize) -> GpuResult<FrameBatch<'g>> {
        let n_frames = unsafe { ffi::vbx_frame_count(samples.len(), bin, hop) };
        Ok(FrameBatch::new(gpu, gpu.upload(samples)?, None, n_frames, bin, hop))
    }

    /// Dense `[F, frame_len]` frames (already windowed by the caller, or rows of another result).
    pub fn dense(gpu: &'g Gpu, frames: &[f64], frame_len: usize) -> GpuResult<FrameBatch<'g>> {
        assert!(frame_len > 0 && frames.len() % frame_len == 0, "frames.len() must be a multiple of frame_len");
        Ok(FrameBatch::new(gpu, gpu.upload(frames)?, None, frames.len() / frame_len, frame_len, frame_len))
    }

    /// Dense rows that already live on the device (e.g. the result of `autocorrelate_rows`).
    pub fn from_device_rows(gpu: &'g Gpu, rows: DeviceBuf<'g, f64>, row_len: usize) -> FrameBatch<'g> {
        let n = rows.len() / row_len;
        FrameBatch::new(gpu, rows, None, n, row_len, row_len)
    }

    pub fn n_frames(&self) -> usize {
        self.n_frames
    }
    pub fn frame_len(&self) -> usize {
        self.frame_len
    }

    /// How many entries of each frame's candidate Vec the drop-in `Pitched::pitch` retrieves.  The default,
    /// `VBX_PITCH_MAX_CANDIDATES(frame_len)`, is the reference's whole Vec (src/periodic.rs:452-454); `1` is the
    /// `PitchExtractor` output (src/periodic.rs:337-353) and an order of magnitude faster (exact top-k pruning).
    pub fn with_pitch_candidates(mut self, kmax: usize) -> FrameBatch<'g> {
        assert!(kmax >= 1 && kmax <= ffi::VBX_MAX_PITCH_CANDIDATES);
        self.pitch_kmax = kmax;
        self
    }

    fn win_ptr(&self) -> *const f64 {
        match &self.window {
            Some(w) => w.as_ptr(),
            None => ptr::null(),
        }
    }

    /// Frame views that implement the reference's traits (drop-in for a user frame loop).
    pub fn frames<'b>(&'b self) -> Frames<'b, 'g> {
        Frames { batch: self, next: 0 }
    }

    pub fn frame<'b>(&'b self, index: usize) -> GpuFrame<'b, 'g> {
        assert!(index < self.n_frames);
        GpuFrame { batch: self, index }
    }

    // ---- periodic.rs ----------------------------------------------------------------------------------------

    /// `Autocorrelate::autocorrelate(n_coeffs)` of every frame (src/periodic.rs:265-289, fold seed quirk
    /// included), left on the device as a dense batch of rows -- ready for `normalize()` and `lpc_all`.
    pub fn autocorrelate_rows(&self, n_coeffs: usize) -> GpuResult<FrameBatch<'g>> {
        let out = self.gpu.alloc::<f64>(self.n_frames * n_coeffs)?;
        self.gpu.check(unsafe {
            ffi::vbx_autocorrelate_f64(self.gpu.raw, self.samples.as_ptr(), self.n_frames, self.frame_len, self.stride, self.win_ptr(), n_coeffs, out.as_mut_ptr())
        })?;
        Ok(FrameBatch::from_device_rows(self.gpu, out, n_coeffs))
    }

    /// The same, copied to the host: `[F, n_coeffs]` row-major.
    pub fn autocorrelate_all(&self, n_coeffs: usize) -> GpuResult<Vec<f64>> {
        self.autocorrelate_rows(n_coeffs)?.samples.to_vec()
    }

    /// `Normalize::normalize` on each row, in place (src/waves.rs:60-76).  Dense, window-less batches only.
    pub fn normalize(&self) -> GpuResult<()> {
        assert!(self.stride == self.frame_len && self.window.is_none(), "normalize() works in place on dense rows");
        self.cache.replace(Cache::default());
        self.gpu.check(unsafe { ffi::vbx_normalize_f64(self.gpu.raw, self.samples.as_mut_ptr(), self.n_frames, self.frame_len) })
    }

    /// `Pitched::pitch::<Hanning>(sample_rate, threshold, _, _, min, max)` of every frame: the first `kmax`
    /// entries of each frame's stable-sorted candidate Vec (`[F, kmax]`, zero padded), the full candidate count
    /// and the status.  Column 0 is the `PitchExtractor` output.
    pub fn pitch_all(&self, sample_rate: f64, threshold: f64, min: f64, max: f64, kmax: usize) -> GpuResult<(Vec<Pitch<f64>>, Vec<i32>, Vec<FrameStatus>)> {
        let (cand, count, status) = self.pitch_raw(sample_rate, threshold, min, max, kmax)?;
        Ok((
            cand.iter().map(|p| Pitch::new(p.frequency, p.strength)).collect(),
            count,
            status.iter().map(|&s| FrameStatus::from_code(s)).collect(),
        ))
    }

    fn pitch_raw(&self, sample_rate: f64, threshold: f64, min: f64, max: f64, kmax: usize) -> GpuResult<(Vec<ffi::VbxPitch>, Vec<i32>, Vec<i32>)> {
        let f = self.n_frames;
        let cand = self.gpu.alloc::<ffi::VbxPitch>(f * kmax)?;
        let count = self.gpu.alloc::<i32>(f)?;
        let status = self.gpu.alloc::<i32>(f)?;
        self.gpu.check(unsafe {
            ffi::vbx_pitch_f64(
                self.gpu.raw, self.samples.as_ptr(), f, self.frame_len, self.stride, self.win_ptr(), sample_rate, threshold, min, max, kmax,
                cand.as_mut_ptr(), count.as_mut_ptr(), status.as_mut_ptr(),
            )
        })?;
        Ok((cand.to_vec()?, count.to_vec()?, status.to_vec()?))
    }

    // ---- spectrum.rs ----------------------------------------------------------------------------------------

    /// `LPC::lpc_mut(n_coeffs, ac, kc, tmp)` with every row of this batch read as an autocorrelation sequence
    /// (src/spectrum.rs:62-84): `(ac [F, n_coeffs + 1], kc [F, n_coeffs])`.  Dense, window-less batches only.
    pub fn lpc_all(&self, n_coeffs: usize) -> GpuResult<(Vec<f64>, Vec<f64>)> {
        assert!(self.window.is_none(), "lpc() reads the receiver as autocorrelation values: build the batch without a window");
        let ac = self.gpu.alloc::<f64>(self.n_frames * (n_coeffs + 1))?;
        let kc = self.gpu.alloc::<f64>(self.n_frames * n_coeffs)?;
        self.gpu.check(unsafe {
            ffi::vbx_lpc_mut_f64(self.gpu.raw, self.samples.as_ptr(), self.n_frames, self.stride, n_coeffs, ac.as_mut_ptr(), kc.as_mut_ptr())
        })?;
        Ok((ac.to_vec()?, kc.to_vec()?))
    }

    /// `frame.autocorrelate(n_coeffs + 1)` [`.normalize()`] `.lpc(n_coeffs)` fused into one pass over the samples
    /// (the `LPCSolver` usage, src/spectrum.rs:40-42): `(r [F, n_coeffs + 1], lpc [F, n_coeffs + 1])`.
    pub fn autocorr_lpc_all(&self, n_coeffs: usize, normalize: bool) -> GpuResult<(Vec<f64>, Vec<f64>)> {
        let r = self.gpu.alloc::<f64>(self.n_frames * (n_coeffs + 1))?;
        let a = self.gpu.alloc::<f64>(self.n_frames * (n_coeffs + 1))?;
        self.gpu.check(unsafe {
            ffi::vbx_autocorr_lpc_f64(
                self.gpu.raw, self.samples.as_ptr(), self.n_frames, self.frame_len, self.stride, self.win_ptr(), n_coeffs, normalize as i32,
                r.as_mut_ptr(), a.as_mut_ptr(),
            )
        })?;
        Ok((r.to_vec()?, a.to_vec()?))
    }

    /// `LPC::lpc_praat(n_coeffs)` of every frame (Burg, src/spectrum.rs:94-146): `[F, n_coeffs]` (no leading 1,
    /// sign-flipped as the reference) and the status (`Lpc` where the reference returns `Err(LPC(..))`).
    pub fn lpc_praat_all(&self, n_coeffs: usize) -> GpuResult<(Vec<f64>, Vec<FrameStatus>)> {
        let (c, s) = self.burg_raw(n_coeffs)?;
        Ok((c, s.iter().map(|&x| FrameStatus::from_code(x)).collect()))
    }

    fn burg_raw(&self, n_coeffs: usize) -> GpuResult<(Vec<f64>, Vec<i32>)> {
        let out = self.gpu.alloc::<f64>(self.n_frames * n_coeffs)?;
        let status = self.gpu.alloc::<i32>(self.n_frames)?;
        self.gpu.check(unsafe {
            ffi::vbx_lpc_burg_f64(
                self.gpu.raw, self.samples.as_ptr(), self.n_frames, self.frame_len, self.stride, self.win_ptr(), n_coeffs, out.as_mut_ptr(),
                status.as_mut_ptr(),
            )
        })?;
        Ok((out.to_vec()?, status.to_vec()?))
    }

    /// `MFCC::mfcc(num_coeffs, freq_bounds, sample_rate)` of every frame (src/spectrum.rs:401-441):
    /// `[F, num_coeffs]` and the status (`Panic` where a mel bin exceeds the spectrum).
    pub fn mfcc_all(&self, num_coeffs: usize, freq_bounds: (f64, f64), sample_rate: f64) -> GpuResult<(Vec<f64>, Vec<FrameStatus>)> {
        let (m, s) = self.mfcc_raw(num_coeffs, freq_bounds, sample_rate)?;
        Ok((m, s.iter().map(|&x| FrameStatus::from_code(x)).collect()))
    }

    fn mfcc_raw(&self, num_coeffs: usize, freq_bounds: (f64, f64), sample_rate: f64) -> GpuResult<(Vec<f64>, Vec<i32>)> {
        let out = self.gpu.alloc::<f64>(self.n_frames * num_coeffs)?;
        let status = self.gpu.alloc::<i32>(self.n_frames)?;
        self.gpu.check(unsafe {
            ffi::vbx_mfcc_f64(
                self.gpu.raw, self.samples.as_ptr(), self.n_frames, self.frame_len, self.stride, self.win_ptr(), num_coeffs, freq_bounds.0,
                freq_bounds.1, sample_rate, out.as_mut_ptr(), status.as_mut_ptr(),
            )
        })?;
        Ok((out.to_vec()?, status.to_vec()?))
    }

    /// `RMS::rms` of every frame (src/waves.rs:10-23).
    pub fn rms_all(&self) -> GpuResult<Vec<f64>> {
        let out = self.gpu.alloc::<f64>(self.n_frames)?;
        self.gpu.check(unsafe {
            ffi::vbx_rms_f64(self.gpu.raw, self.samples.as_ptr(), self.n_frames, self.frame_len, self.stride, self.win_ptr(), out.as_mut_ptr())
        })?;
        out.to_vec()
    }
}

// ---------------------------------------------------------------------------------------------------------------
// drop-in frame views: the reference's traits, one frame at a time, computed for the whole batch on first use
// ---------------------------------------------------------------------------------------------------------------

/// Iterator over the frame views of a batch (`FrameBatch::frames`).
pub struct Frames<'b, 'g: 'b> {
    batch: &'b FrameBatch<'g>,
    next: usize,
}

impl<'b, 'g: 'b> Iterator for Frames<'b, 'g> {
    type Item = GpuFrame<'b, 'g>;
    fn next(&mut self) -> Option<GpuFrame<'b, 'g>> {
        if self.next >= self.batch.n_frames {
            return None;
        }
        self.next += 1;
        Some(GpuFrame { batch: self.batch, index: self.next - 1 })
    }
}

/// Frame `index` of a [`FrameBatch`].  Implements `Autocorrelate<f64>`, `LPC<f64>`, `Pitched<f64, f64>` and
/// `MFCC<f64>` exactly as `[f64]` does in the reference; the first call of a method (per parameter set) runs the
/// whole batch on the GPU, later calls copy a row of the cached result.
pub struct GpuFrame<'b, 'g: 'b> {
    batch: &'b FrameBatch<'g>,
    index: usize,
}

impl<'b, 'g: 'b> GpuFrame<'b, 'g> {
    pub fn index(&self) -> usize {
        self.index
    }
}

fn expect_gpu<T>(r: GpuResult<T>) -> T {
    match r {
        Ok(v) => v,
        Err(e) => panic!("{}", e),
    }
}

/// src/periodic.rs:265-274
impl<'b, 'g: 'b> Autocorrelate<f64> for GpuFrame<'b, 'g> {
    fn autocorrelate_mut(&self, coeffs: &mut [f64]) {
        let n = coeffs.len();
        let mut cache = self.batch.cache.borrow_mut();
        let stale = match &cache.autocorr {
            Some((lags, _)) => *lags != n,
            None => true,
        };
        if stale {
            cache.autocorr = Some((n, expect_gpu(self.batch.autocorrelate_all(n))));
        }
        let rows = &cache.autocorr.as_ref().unwrap().1;
        coeffs.copy_from_slice(&rows[self.index * n..(self.index + 1) * n]);
    }
}

/// src/spectrum.rs:50-55.  As in the reference, `lpc` / `lpc_mut` read the receiver as an AUTOCORRELATION sequence
/// (call them on the frames of `batch.autocorrelate_rows(n)`), `lpc_praat` / `lpc_praat_mut` as samples.
impl<'b, 'g: 'b> LPC<f64> for GpuFrame<'b, 'g> {
    /// `tmp` is scratch in the reference (its final content is a stale copy of `ac`); it is left untouched here.
    fn lpc_mut(&self, n_coeffs: usize, ac: &mut [f64], kc: &mut [f64], _tmp: &mut [f64]) {
        let mut cache = self.batch.cache.borrow_mut();
        let stale = match &cache.lpc {
            Some((p, _, _)) => *p != n_coeffs,
            None => true,
        };
        if stale {
            let (a, k) = expect_gpu(self.batch.lpc_all(n_coeffs));
            cache.lpc = Some((n_coeffs, a, k));
        }
        let (_, a, k) = cache.lpc.as_ref().unwrap();
        ac[..n_coeffs + 1].copy_from_slice(&a[self.index * (n_coeffs + 1)..(self.index + 1) * (n_coeffs + 1)]);
        kc[..n_coeffs].copy_from_slice(&k[self.index * n_coeffs..(self.index + 1) * n_coeffs]);
    }

    fn lpc(&self, n_coeffs: usize) -> Vec<f64> {
        let mut ac = vec![0f64; n_coeffs + 1];
        let mut kc = vec![0f64; n_coeffs];
        let mut tmp = vec![0f64; n_coeffs];
        self.lpc_mut(n_coeffs, &mut ac[..], &mut kc[..], &mut tmp[..]);
        ac
    }

    /// `work` is the reference's `2 * len + n_coeffs` scratch (asserted, unused: the library owns its workspaces).
    fn lpc_praat_mut(&self, n_coeffs: usize, coeffs: &mut [f64], work: &mut [f64]) -> VoxBoxResult<()> {
        assert!(coeffs.len() >= n_coeffs);
        assert!(work.len() >= self.batch.frame_len * 2 + n_coeffs);
        let mut cache = self.batch.cache.borrow_mut();
        let stale = match &cache.burg {
            Some((p, _, _)) => *p != n_coeffs,
            None => true,
        };
        if stale {
            let (c, s) = expect_gpu(self.batch.burg_raw(n_coeffs));
            cache.burg = Some((n_coeffs, c, s));
        }
        let (_, c, s) = cache.burg.as_ref().unwrap();
        FrameStatus::from_code(s[self.index]).into_result()?;
        coeffs[..n_coeffs].copy_from_slice(&c[self.index * n_coeffs..(self.index + 1) * n_coeffs]);
        Ok(())
    }

    fn lpc_praat(&self, n_coeffs: usize) -> VoxBoxResult<Vec<f64>> {
        let mut coeffs = vec![0f64; n_coeffs];
        let mut work = vec![0f64; self.batch.frame_len * 2 + n_coeffs];
        self.lpc_praat_mut(n_coeffs, &mut coeffs[..], &mut work[..])?;
        Ok(coeffs)
    }
}

/// src/periodic.rs:356-358.  `W` can only be `Hanning` (the reference's one `LagType`); the library divides by the
/// `HanningLag` table.  `local_peak` / `global_peak` are ignored, as in the reference (src/periodic.rs:396).
impl<'b, 'g: 'b> Pitched<f64, f64> for GpuFrame<'b, 'g> {
    fn pitch<W: LagType>(&self, sample_rate: f64, threshold: f64, _local_peak: f64, _global_peak: f64, min: f64, max: f64) -> Vec<Pitch<f64>> {
        let kmax = self.batch.pitch_kmax;
        let key = PitchKey { sample_rate: sample_rate.to_bits(), threshold: threshold.to_bits(), min: min.to_bits(), max: max.to_bits(), kmax };
        let mut cache = self.batch.cache.borrow_mut();
        let stale = match &cache.pitch {
            Some(rows) => rows.key != key,
            None => true,
        };
        if stale {
            let (cand, count, status) = expect_gpu(self.batch.pitch_raw(sample_rate, threshold, min, max, kmax));
            cache.pitch = Some(PitchRows { key, cand, count, status });
        }
        let rows = cache.pitch.as_ref().unwrap();
        FrameStatus::from_code(rows.status[self.index]).unwrap_or_panic();
        let n = (rows.count[self.index] as usize).min(kmax);
        rows.cand[self.index * kmax..self.index * kmax + n].iter().map(|p| Pitch::new(p.frequency, p.strength)).collect()
    }
}

/// src/spectrum.rs:371-373
impl<'b, 'g: 'b> MFCC<f64> for GpuFrame<'b, 'g> {
    fn mfcc(&self, num_coeffs: usize, freq_bounds: (f64, f64), sample_rate: f64) -> Vec<f64> {
        let key = (num_coeffs, freq_bounds.0.to_bits(), freq_bounds.1.to_bits(), sample_rate.to_bits());
        let mut cache = self.batch.cache.borrow_mut();
        let stale = match &cache.mfcc {
            Some((k, _, _)) => *k != key,
            None => true,
        };
        if stale {
            let (m, s) = expect_gpu(self.batch.mfcc_raw(num_coeffs, freq_bounds, sample_rate));
            cache.mfcc = Some((key, m, s));
        }
        let (_, m, s) = cache.mfcc.as_ref().unwrap();
        FrameStatus::from_code(s[self.index]).unwrap_or_panic();
        m[self.index * num_coeffs..(self.index + 1) * num_coeffs].to_vec()
    }
}

// ---------------------------------------------------------------------------------------------------------------
// polynomial.rs / ToResonance
// ---------------------------------------------------------------------------------------------------------------

/// `F` polynomials of `len` coefficients each (coefficient of x^j at index j) in HBM: the batched receiver of
/// `Polynomial::{find_roots, laguerre, div_polynomial}` (src/polynomial.rs:10-21).
pub struct PolyBatch<'g> {
    gpu: &'g Gpu,
    polys: DeviceBuf<'g, Complex<f64>>,
    n_polys: usize,
    len: usize,
}

impl<'g> PolyBatch<'g> {
    pub fn new(gpu: &'g Gpu, polys: &[Complex<f64>], len: usize) -> GpuResult<PolyBatch<'g>> {
        assert!(len > 0 && polys.len() % len == 0);
        Ok(PolyBatch { gpu, polys: gpu.upload(polys)?, n_polys: polys.len() / len, len })
    }

    /// `Polynomial::find_roots` of every polynomial (src/polynomial.rs:79-152: Laguerre from -2-2i with deflation,
    /// roots in discovery order, remainder zeroed): the root rows, ready for `to_resonance`, and the status.
    /// Consumes the batch: `find_roots_mut` overwrites the coefficients with the roots, as in the reference.
    pub fn find_roots(self) -> GpuResult<(RootRows<'g>, Vec<FrameStatus>)> {
        let status = self.gpu.alloc::<i32>(self.n_polys)?;
        self.gpu.check(unsafe {
            ffi::vbx_find_roots_c64(self.gpu.raw, self.polys.as_mut_ptr() as *mut ffi::VbxComplex, self.n_polys, self.len, status.as_mut_ptr())
        })?;
        let st = status.to_vec()?.iter().map(|&s| FrameStatus::from_code(s)).collect();
        Ok((RootRows { gpu: self.gpu, roots: self.polys, n_rows: self.n_polys, n_roots: self.len, cache: RefCell::new(None) }, st))
    }

    /// `Polynomial::laguerre(start)` of every polynomial (src/polynomial.rs:34-72).
    pub fn laguerre(&self, start: Complex<f64>) -> GpuResult<Vec<Complex<f64>>> {
        let out = self.gpu.alloc::<Complex<f64>>(self.n_polys)?;
        self.gpu.check(unsafe {
            ffi::vbx_laguerre_c64(
                self.gpu.raw, self.polys.as_ptr() as *const ffi::VbxComplex, self.n_polys, self.len, ffi::VbxComplex { re: start.re, im: start.im },
                out.as_mut_ptr() as *mut ffi::VbxComplex,
            )
        })?;
        out.to_vec()
    }

    /// `Polynomial::div_polynomial_mut(other, rem)` of every polynomial by `(x + others[f])` (src/polynomial.rs:155-195):
    /// the quotients replace the batch, the remainders are returned with the status (`Polynomial` where `other` is 0).
    pub fn div_polynomial(&mut self, others: &[Complex<f64>]) -> GpuResult<(Vec<Complex<f64>>, Vec<FrameStatus>)> {
        assert_eq!(others.len(), self.n_polys);
        let d_others = self.gpu.upload(others)?;
        let rem = self.gpu.alloc::<Complex<f64>>(self.n_polys * self.len)?;
        let status = self.gpu.alloc::<i32>(self.n_polys)?;
        self.gpu.check(unsafe {
            ffi::vbx_div_polynomial_c64(
                self.gpu.raw, self.polys.as_mut_ptr() as *mut ffi::VbxComplex, d_others.as_ptr() as *const ffi::VbxComplex, self.n_polys, self.len,
                rem.as_mut_ptr() as *mut ffi::VbxComplex, status.as_mut_ptr(),
            )
        })?;
        Ok((rem.to_vec()?, status.to_vec()?.iter().map(|&s| FrameStatus::from_code(s)).collect()))
    }

    /// The polynomials (or, after `div_polynomial`, the quotients) on the host.
    pub fn to_vec(&self) -> GpuResult<Vec<Complex<f64>>> {
        self.polys.to_vec()
    }
}

/// `[F, n_roots]` complex roots in HBM: the batched receiver of `ToResonance::to_resonance`.
pub struct RootRows<'g> {
    gpu: &'g Gpu,
    roots: DeviceBuf<'g, Complex<f64>>,
    n_rows: usize,
    n_roots: usize,
    cache: RefCell<Option<(u64, Vec<Resonance<f64>>, Vec<i32>)>>,
}

impl<'g> RootRows<'g> {
    pub fn new(gpu: &'g Gpu, roots: &[Complex<f64>], n_roots: usize) -> GpuResult<RootRows<'g>> {
        assert!(n_roots > 0 && roots.len() % n_roots == 0);
        Ok(RootRows { gpu, roots: gpu.upload(roots)?, n_rows: roots.len() / n_roots, n_roots, cache: RefCell::new(None) })
    }

    pub fn to_vec(&self) -> GpuResult<Vec<Complex<f64>>> {
        self.roots.to_vec()
    }

    /// `ToResonance::to_resonance(sample_rate)` of every row (src/spectrum.rs:165-210): `[F, n_roots]` resonances
    /// sorted by frequency and zero padded, and the number of resonances of each row.
    pub fn to_resonance_all(&self, sample_rate: f64) -> GpuResult<(Vec<Resonance<f64>>, Vec<i32>)> {
        // Resonance<f64> is #[repr(C)] { frequency, bandwidth } (src/spectrum.rs:149-154) == vbx_resonance
        let out = self.gpu.alloc::<Resonance<f64>>(self.n_rows * self.n_roots)?;
        let count = self.gpu.alloc::<i32>(self.n_rows)?;
        self.gpu.check(unsafe {
            ffi::vbx_to_resonance_c64(
                self.gpu.raw, self.roots.as_ptr() as *const ffi::VbxComplex, self.n_rows, self.n_roots, sample_rate,
                out.as_mut_ptr() as *mut ffi::VbxResonance, count.as_mut_ptr(),
            )
        })?;
        Ok((out.to_vec()?, count.to_vec()?))
    }

    pub fn row<'b>(&'b self, index: usize) -> RootRow<'b, 'g> {
        assert!(index < self.n_rows);
        RootRow { rows: self, index }
    }
}

/// One row of roots; implements `ToResonance<f64>` (src/spectrum.rs:195-210) by reading its row of the batch result.
pub struct RootRow<'b, 'g: 'b> {
    rows: &'b RootRows<'g>,
    index: usize,
}

impl<'b, 'g: 'b> ToResonance<f64> for RootRow<'b, 'g> {
    fn to_resonance(&self, sample_rate: f64) -> Vec<Resonance<f64>> {
        let mut cache = self.rows.cache.borrow_mut();
        let stale = match &*cache {
            Some((sr, _, _)) => *sr != sample_rate.to_bits(),
            None => true,
        };
        if stale {
            let (res, count) = expect_gpu(self.rows.to_resonance_all(sample_rate));
            *cache = Some((sample_rate.to_bits(), res, count));
        }
        let (_, res, count) = cache.as_ref().unwrap();
        let n = self.rows.n_roots;
        res[self.index * n..self.index * n + count[self.index] as usize].to_vec()
    }
}

// ---------------------------------------------------------------------------------------------------------------
// lib.rs: find_formants
// ---------------------------------------------------------------------------------------------------------------

/// `vox_box::find_formants(buf, sample_rate, resample_ratio, resampled_buf, n_coeffs, work, complex_work, formants)`
/// (src/lib.rs:40-116) over every frame of `frames` (rectangular frames: the periodic Hanning of src/lib.rs:65-70
/// is applied inside), with the caller's loop-carried `formants` state (tests/lib.rs:75-79) carried on the device.
///
/// * `seg_start`: ascending frame indices at which the state is reset to the incoming `formants` (utterance
///   boundaries; `&[0]` or `&[]` = one utterance);
/// * `formants`: in = the initial estimates (`MALE_FORMANT_ESTIMATES` ..), out = the state after the last frame;
/// * returns `(track [F, formants.len()], status [F])`: the estimates after each frame.  A frame whose status is
///   not `Ok` leaves the state untouched, as `?` does at src/lib.rs:75.
///
/// `resample_ratio != 1.0` runs the linear resampler of src/lib.rs:57-61 first (parity unpinned: the arithmetic
/// lives in the un-vendored `sample 0.10` crate).  The reference's `resampled_buf` / `work` / `complex_work`
/// arguments have no counterpart: the library owns its workspaces.
pub fn find_formants(frames: &FrameBatch, sample_rate: f64, resample_ratio: f64, n_coeffs: usize, seg_start: &[i64], formants: &mut [Resonance<f64>]) -> VoxBoxResult<(Vec<Resonance<f64>>, Vec<FrameStatus>)> {
    let gpu = frames.gpu;
    let f = frames.n_frames;
    let n_est = formants.len();
    let gpu_err = |_e: GpuError| VoxBoxError::Workspace; // the reference's only non-algorithmic error (src/lib.rs:46-48)
    assert!(frames.window.is_none(), "find_formants applies its own window: pass rectangular frames (tests/lib.rs:71)");

    // resample front end (src/lib.rs:42,57-61)
    let resampled;
    let (x, frame_len, stride) = if resample_ratio != 1.0 {
        let m = unsafe { ffi::vbx_resampled_len(frames.frame_len, resample_ratio) };
        resampled = gpu.alloc::<f64>(f * m).map_err(gpu_err)?;
        gpu.check(unsafe {
            ffi::vbx_resample_linear_f64(gpu.raw, frames.samples.as_ptr(), f, frames.frame_len, frames.stride, resample_ratio, resampled.as_mut_ptr())
        })
        .map_err(gpu_err)?;
        (resampled.as_ptr(), m, m)
    } else {
        (frames.samples.as_ptr(), frames.frame_len, frames.stride)
    };

    let track = gpu.alloc::<Resonance<f64>>(f * n_est).map_err(gpu_err)?;
    let status = gpu.alloc::<i32>(f).map_err(gpu_err)?;
    let (seg_ptr, n_seg) = if seg_start.is_empty() { (ptr::null(), 0) } else { (seg_start.as_ptr(), seg_start.len()) };
    gpu.check(unsafe {
        ffi::vbx_find_formants_f64(
            gpu.raw, x, f, frame_len, stride, sample_rate, n_coeffs, seg_ptr, n_seg, formants.as_ptr() as *const ffi::VbxResonance, n_est,
            track.as_mut_ptr() as *mut ffi::VbxResonance, ptr::null_mut(), ptr::null_mut(), ptr::null_mut(), status.as_mut_ptr(),
        )
    })
    .map_err(gpu_err)?;
    let track = track.to_vec().map_err(gpu_err)?;
    let status: Vec<FrameStatus> = status.to_vec().map_err(gpu_err)?.iter().map(|&s| FrameStatus::from_code(s)).collect();
    if f > 0 {
        formants.copy_from_slice(&track[(f - 1) * n_est..f * n_est]);
    }
    Ok((track, status))
}

// ---------------------------------------------------------------------------------------------------------------
// spectrum.rs: EstimateFormants / FormantExtractor over resonance rows on the device
// ---------------------------------------------------------------------------------------------------------------

/// Resonance rows `[F, n_res]` on the device (zero padded, sorted by frequency: what `find_formants` builds at
/// src/lib.rs:94-110, or whatever the caller uploads).  The receiver of the batched `EstimateFormants`.
pub struct ResonanceRows<'g> {
    gpu: &'g Gpu,
    rows: DeviceBuf<'g, Resonance<f64>>,
    n_frames: usize,
    n_res: usize,
}

impl<'g> ResonanceRows<'g> {
    /// `rows`: `n_frames * n_res` resonances, row-major.
    pub fn new(gpu: &'g Gpu, rows: &[Resonance<f64>], n_res: usize) -> GpuResult<ResonanceRows<'g>> {
        assert!(n_res >= 1 && rows.len() % n_res == 0, "rows must hold whole rows of n_res resonances");
        Ok(ResonanceRows { gpu, rows: gpu.upload(rows)?, n_frames: rows.len() / n_res, n_res })
    }

    pub fn n_frames(&self) -> usize {
        self.n_frames
    }

    /// `EstimateFormants::estimate_formants` (src/spectrum.rs:232-333) applied frame after frame, the estimates carried
    /// from each frame to the next exactly as `FormantExtractor::next` does (src/spectrum.rs:357-369): returns the
    /// estimates after every frame, `[F, estimates.len()]`, and leaves the state after the last frame in `estimates`.
    /// `seg_start`: frames at which the state restarts from the incoming estimates (`&[]` = one utterance);
    /// `frame_status`: frames whose status is not `Ok` leave the state untouched (src/lib.rs:75 `?`).
    /// Utterances of a few hundred frames or more take the library's chunked scan (bit-identical to the sequential one).
    pub fn estimate_formants_all(&self, estimates: &mut [Resonance<f64>], seg_start: &[i64], frame_status: Option<&[i32]>) -> GpuResult<Vec<Resonance<f64>>> {
        let gpu = self.gpu;
        let n_est = estimates.len();
        let out = gpu.alloc::<Resonance<f64>>(self.n_frames * n_est)?;
        let st_dev = match frame_status {
            Some(s) => {
                assert_eq!(s.len(), self.n_frames);
                Some(gpu.upload(s)?)
            }
            None => None,
        };
        let (seg_ptr, n_seg) = if seg_start.is_empty() { (ptr::null(), 0) } else { (seg_start.as_ptr(), seg_start.len()) };
        gpu.check(unsafe {
            ffi::vbx_estimate_formants_f64(
                gpu.raw, self.rows.as_ptr() as *const ffi::VbxResonance, self.n_frames, self.n_res, seg_ptr, n_seg,
                estimates.as_ptr() as *const ffi::VbxResonance, n_est, st_dev.as_ref().map_or(ptr::null(), |d| d.as_ptr()),
                out.as_mut_ptr() as *mut ffi::VbxResonance,
            )
        })?;
        let track = out.to_vec()?;
        if self.n_frames > 0 {
            estimates.copy_from_slice(&track[(self.n_frames - 1) * n_est..self.n_frames * n_est]);
        }
        Ok(track)
    }

    /// The reference's `FormantExtractor::new(num_formants, resonances, starting_estimates)` (src/spectrum.rs:343-355)
    /// with the whole scan done up front on the device: an iterator over the per-frame estimate Vecs.
    pub fn formant_extractor(&self, num_formants: usize, starting_estimates: Vec<Resonance<f64>>) -> GpuResult<FormantExtractor> {
        assert_eq!(num_formants, starting_estimates.len(), "one starting estimate per formant");
        let mut est = starting_estimates;
        let track = self.estimate_formants_all(&mut est, &[], None)?;
        Ok(FormantExtractor { estimates: est, num_formants, track, next: 0 })
    }
}

/// `vox_box::spectrum::FormantExtractor` (src/spectrum.rs:336-369) over a scan the device has already done:
/// `next()` yields the estimates after each frame; `estimates` holds the state after the last one.
pub struct FormantExtractor {
    pub estimates: Vec<Resonance<f64>>,
    num_formants: usize,
    track: Vec<Resonance<f64>>,
    next: usize,
}

impl Iterator for FormantExtractor {
    type Item = Vec<Resonance<f64>>;

    fn next(&mut self) -> Option<Vec<Resonance<f64>>> {
        let n = self.num_formants;
        if n == 0 || (self.next + 1) * n > self.track.len() {
            return None;
        }
        let row = self.track[self.next * n..(self.next + 1) * n].to_vec();
        self.next += 1;
        Some(row)
    }
}

/// One frame's `EstimateFormants<f64>` (src/spectrum.rs:216-219) through the library: `self` = the estimates (in/out),
/// `resonances` = the frame's row.  For code that keeps the reference's per-frame loop; a batch should call
/// [`ResonanceRows::estimate_formants_all`] once instead.
pub struct GpuEstimates<'g> {
    pub gpu: &'g Gpu,
    pub estimates: Vec<Resonance<f64>>,
}

impl<'g> vox_box::spectrum::EstimateFormants<f64> for GpuEstimates<'g> {
    type FormantSlots = [Option<Resonance<f64>>; 6];

    fn estimate_formants(&mut self, resonances: &[Resonance<f64>]) {
        let rows = expect_gpu(ResonanceRows::new(self.gpu, resonances, resonances.len().max(1)));
        let mut est = self.estimates.clone();
        expect_gpu(rows.estimate_formants_all(&mut est, &[], None));
        self.estimates = est;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the user's frame loop, fused: vbx_analyze_frames_f64 / _pcm16
// ---------------------------------------------------------------------------------------------------------------

/// What one call of the fused frame loop computes per frame (`vbx_analysis_params`): the parts with order / count 0 are
/// skipped.  `Default` = the bench's pipeline: pitch (0.2, 75..600 Hz), LPC(12), find_formants(12) from the male
/// estimates (src/lib.rs:27), MFCC(13, 100..8000 Hz).
#[derive(Clone, Debug)]
pub struct AnalysisParams {
    pub sample_rate: f64,
    pub pitch: (f64, f64, f64),
    pub lpc_order: usize,
    pub formant_order: usize,
    pub est_init: Vec<Resonance<f64>>,
    pub mfcc: Option<(usize, f64, f64)>,
}

impl AnalysisParams {
    pub fn new(sample_rate: f64) -> AnalysisParams {
        let est_init = ffi_male_estimates();
        AnalysisParams { sample_rate, pitch: (0.2, 75.0, 600.0), lpc_order: 12, formant_order: 12, est_init, mfcc: Some((13, 100.0, 8000.0)) }
    }

    fn to_ffi(&self) -> ffi::VbxAnalysisParams {
        assert!(self.est_init.len() <= ffi::VBX_FORMANT_SLOTS, "at most 6 formant slots (src/spectrum.rs:228)");
        let mut est = [ffi::VbxResonance { frequency: 0.0, bandwidth: 0.0 }; ffi::VBX_FORMANT_SLOTS];
        for (slot, r) in est.iter_mut().zip(self.est_init.iter()) {
            *slot = ffi::VbxResonance { frequency: r.frequency, bandwidth: r.bandwidth };
        }
        let (k, lo, hi) = self.mfcc.unwrap_or((0, 0.0, 0.0));
        ffi::VbxAnalysisParams {
            sample_rate: self.sample_rate,
            pitch_threshold: self.pitch.0,
            pitch_fmin: self.pitch.1,
            pitch_fmax: self.pitch.2,
            lpc_order: self.lpc_order,
            formant_order: self.formant_order,
            n_est: if self.formant_order > 0 { self.est_init.len() } else { 0 },
            est_init: est,
            mfcc_coeffs: k,
            mfcc_lo_hz: lo,
            mfcc_hi_hz: hi,
        }
    }

    /// Doubles per record (`vbx_record_doubles`) and the first column of each part: (pitch, formants, mfcc, lpc).
    pub fn layout(&self) -> (usize, usize, usize, usize, usize) {
        let p = self.to_ffi();
        let rec = unsafe { ffi::vbx_record_doubles(&p) };
        let c_form = 2;
        let c_mfcc = c_form + 2 * p.n_est;
        let c_lpc = c_mfcc + p.mfcc_coeffs;
        (rec, 0, c_form, c_mfcc, c_lpc)
    }
}

fn ffi_male_estimates() -> Vec<Resonance<f64>> {
    vox_box::MALE_FORMANT_ESTIMATES.iter().map(|&f| Resonance::new(f, 1.0)).collect()
}

/// The records of one fused call: `[F, record_ld]` doubles on the device (the buffer the multi-GPU gather sends) plus
/// the three status rows.
pub struct Records<'g> {
    pub data: DeviceBuf<'g, f64>,
    pub status3: DeviceBuf<'g, i32>,
    pub n_frames: usize,
    pub record_ld: usize,
}

impl<'g> Records<'g> {
    /// `(records [F * record_ld], pitch status [F], formant status [F], mfcc status [F])` on the host.
    pub fn to_host(&self) -> GpuResult<(Vec<f64>, Vec<FrameStatus>, Vec<FrameStatus>, Vec<FrameStatus>)> {
        let st = self.status3.to_vec()?;
        let f = self.n_frames;
        let conv = |s: &[i32]| s.iter().map(|&c| FrameStatus::from_code(c)).collect::<Vec<_>>();
        Ok((self.data.to_vec()?, conv(&st[0..f]), conv(&st[f..2 * f]), conv(&st[2 * f..3 * f])))
    }

    /// One-device form of [`Comm::stitch_tracks`] (`vbx_track_stitch_f64`): these records -- the LAST `analyze` call on their
    /// `Gpu` -- hold a shard whose tracker started from a guess; `state` holds the true state before row `first` (the
    /// `n_est` resonances of the formant row the previous shard ends with).  Rows `[first, stop)` are corrected where
    /// they differ; the result is the sequential scan's, bit for bit.
    pub fn stitch_tracks_from(&self, first: usize, stop: usize, state: &DeviceBuf<f64>, n_est: usize) -> GpuResult<()> {
        assert!(first <= stop && stop <= self.n_frames);
        assert!(n_est >= 1 && state.len() >= 2 * n_est && self.record_ld >= 2 + 2 * n_est);
        let gpu = self.data.gpu;
        let formants = unsafe { self.data.as_mut_ptr().add(2) } as *mut ffi::VbxResonance;
        gpu.check(unsafe {
            ffi::vbx_track_stitch_f64(gpu.raw, formants, self.n_frames, self.record_ld, first, stop, state.as_ptr() as *const ffi::VbxResonance, ptr::null_mut())
        })
    }
}

impl<'g> FrameBatch<'g> {
    /// The user's whole frame loop in one call (`vbx_analyze_frames_f64`): per frame
    /// `pitch::<Hanning>(..)[0]`, `autocorrelate(p + 1)` -> `lpc(p)`, `find_formants(..)` with the state carried per
    /// utterance, `mfcc(..)` -- examples/pitch_detection.rs:23-30, tests/lib.rs:71-83 -- as one fixed-size record per
    /// frame.  The batch must be a rectangular view (the library applies the Hanning window of the `Windower` and the
    /// periodic Hanning of `find_formants` itself).
    pub fn analyze(&self, params: &AnalysisParams, seg_start: &[i64]) -> GpuResult<Records<'g>> {
        assert!(self.window.is_none(), "analyze applies the windows itself: pass a rectangular view (windower_rectangle)");
        let gpu = self.gpu;
        let p = params.to_ffi();
        let rec = unsafe { ffi::vbx_record_doubles(&p) };
        let ld = rec + (rec & 1);
        let data = gpu.alloc::<f64>(self.n_frames * ld)?;
        let status3 = gpu.alloc::<i32>(3 * self.n_frames)?;
        let (seg_ptr, n_seg) = if seg_start.is_empty() { (ptr::null(), 0) } else { (seg_start.as_ptr(), seg_start.len()) };
        gpu.check(unsafe {
            ffi::vbx_analyze_frames_f64(gpu.raw, self.samples.as_ptr(), self.n_frames, self.frame_len, self.stride, &p, seg_ptr, n_seg,
                                        data.as_mut_ptr(), ld, status3.as_mut_ptr())
        })?;
        Ok(Records { data, status3, n_frames: self.n_frames, record_ld: ld })
    }
}

/// 16-bit PCM samples on the device with a `Windower` view over them: what a WAV reader hands the reference's callers
/// before the `as f64 / 32767` of tests/lib.rs:17-19.  A quarter of the bytes of the f64 view: the form for host-fed
/// operation.
pub struct PcmBatch<'g> {
    gpu: &'g Gpu,
    samples: DeviceBuf<'g, i16>,
    n_frames: usize,
    frame_len: usize,
    stride: usize,
}

impl<'g> PcmBatch<'g> {
    /// `Windower::rectangle(samples, bin, hop)` semantics (frame t = samples[t*hop .. t*hop+bin] while bin <= remaining).
    pub fn windower(gpu: &'g Gpu, samples: &[i16], bin: usize, hop: usize) -> GpuResult<PcmBatch<'g>> {
        let n_frames = unsafe { ffi::vbx_frame_count(samples.len(), bin, hop) };
        Ok(PcmBatch { gpu, samples: gpu.upload(samples)?, n_frames, frame_len: bin, stride: hop })
    }

    pub fn n_frames(&self) -> usize {
        self.n_frames
    }

    /// `samples as f64 / 32767` for every sample, on the device (`vbx_pcm16_to_f64`): the f64 batch of the same view.
    pub fn widen(&self) -> GpuResult<FrameBatch<'g>> {
        let out = self.gpu.alloc::<f64>(self.samples.len())?;
        self.gpu.check(unsafe { ffi::vbx_pcm16_to_f64(self.gpu.raw, self.samples.as_ptr(), self.samples.len(), out.as_mut_ptr()) })?;
        Ok(FrameBatch::new(self.gpu, out, None, self.n_frames, self.frame_len, self.stride))
    }

    /// [`FrameBatch::analyze`] reading the PCM directly (`vbx_analyze_frames_pcm16`): bit-identical records to
    /// `self.widen()?.analyze(..)`.
    pub fn analyze(&self, params: &AnalysisParams, seg_start: &[i64]) -> GpuResult<Records<'g>> {
        let gpu = self.gpu;
        let p = params.to_ffi();
        let rec = unsafe { ffi::vbx_record_doubles(&p) };
        let ld = rec + (rec & 1);
        let data = gpu.alloc::<f64>(self.n_frames * ld)?;
        let status3 = gpu.alloc::<i32>(3 * self.n_frames)?;
        let (seg_ptr, n_seg) = if seg_start.is_empty() { (ptr::null(), 0) } else { (seg_start.as_ptr(), seg_start.len()) };
        gpu.check(unsafe {
            ffi::vbx_analyze_frames_pcm16(gpu.raw, self.samples.as_ptr(), self.n_frames, self.frame_len, self.stride, &p, seg_ptr, n_seg,
                                          data.as_mut_ptr(), ld, status3.as_mut_ptr())
        })?;
        Ok(Records { data, status3, n_frames: self.n_frames, record_ld: ld })
    }
}

// ---------------------------------------------------------------------------------------------------------------
// multi-GPU: frame-range sharding and the record gather (one process per GPU, RCCL inside the library)
// ---------------------------------------------------------------------------------------------------------------

/// Frames `[lo, hi)` of `rank` out of `world` (`vbx_shard_range`): the even split; with `seg_start` a cut moves up to an
/// utterance start within 1/32 of a shard after it.  A cut inside an utterance is carried across by [`shard_plan`] +
/// [`Comm::stitch_tracks`].
pub fn shard_range(n_frames: usize, world: i32, rank: i32, seg_start: &[i64]) -> GpuResult<(usize, usize)> {
    let (mut lo, mut hi) = (0usize, 0usize);
    let (seg_ptr, n_seg) = if seg_start.is_empty() { (ptr::null(), 0) } else { (seg_start.as_ptr(), seg_start.len()) };
    let rc = unsafe { ffi::vbx_shard_range(n_frames, world, rank, seg_ptr, n_seg, &mut lo, &mut hi) };
    if rc != ffi::VBX_SUCCESS {
        return Err(GpuError { code: rc, message: last_error(ptr::null()) });
    }
    Ok((lo, hi))
}

/// The mel filter bank's bins `floor((N + 1) hz / sr)` at `num_coeffs + 2` mel-spaced points (`vbx_mfcc_bins`,
/// src/spectrum.rs:411-414) and whether the reference panics on every frame of this geometry (a bin beyond the spectrum).
pub fn mfcc_bins(frame_len: usize, num_coeffs: usize, freq_bounds: (f64, f64), sample_rate: f64) -> GpuResult<(Vec<i32>, bool)> {
    let mut bins = vec![0i32; num_coeffs + 2];
    let rc = unsafe { ffi::vbx_mfcc_bins(frame_len, num_coeffs, freq_bounds.0, freq_bounds.1, sample_rate, bins.as_mut_ptr()) };
    if rc < 0 {
        return Err(GpuError { code: rc, message: last_error(ptr::null()) });
    }
    Ok((bins, rc == 1))
}

/// One rank's part of a recording sharded by frame ranges (`vbx_shard_plan`): its frames, the warm-up frames before them
/// and whether its formant track continues from the previous rank / into the next one.
pub fn shard_plan(n_frames: usize, world: i32, rank: i32, seg_start: &[i64]) -> GpuResult<ffi::VbxShardPlan> {
    let mut plan = ffi::VbxShardPlan::default();
    let (seg_ptr, n_seg) = if seg_start.is_empty() { (ptr::null(), 0) } else { (seg_start.as_ptr(), seg_start.len()) };
    let rc = unsafe { ffi::vbx_shard_plan(n_frames, world, rank, seg_ptr, n_seg, &mut plan) };
    if rc != ffi::VBX_SUCCESS {
        return Err(GpuError { code: rc, message: last_error(ptr::null()) });
    }
    Ok(plan)
}

/// Utterance starts of the frames `[lo - warm, hi)` of a plan, re-based to the shard (`vbx_shard_local_segments`): the
/// `seg_start` of the rank's own `analyze` / `find_formants` call.
pub fn shard_local_segments(plan: &ffi::VbxShardPlan, seg_start: &[i64]) -> GpuResult<Vec<i64>> {
    let (seg_ptr, n_seg) = if seg_start.is_empty() { (ptr::null(), 0) } else { (seg_start.as_ptr(), seg_start.len()) };
    let mut n = 0usize;
    let rc = unsafe { ffi::vbx_shard_local_segments(plan, seg_ptr, n_seg, ptr::null_mut(), 0, &mut n) };
    if rc != ffi::VBX_SUCCESS {
        return Err(GpuError { code: rc, message: last_error(ptr::null()) });
    }
    let mut out = vec![0i64; n];
    let rc = unsafe { ffi::vbx_shard_local_segments(plan, seg_ptr, n_seg, out.as_mut_ptr(), out.len(), &mut n) };
    if rc != ffi::VBX_SUCCESS {
        return Err(GpuError { code: rc, message: last_error(ptr::null()) });
    }
    Ok(out)
}

/// Samples `[s0, s1)` the frames `[lo, hi)` read, including the `frame_len - hop` halo (`vbx_shard_samples`).
pub fn shard_samples(lo: usize, hi: usize, frame_len: usize, hop: usize) -> GpuResult<(usize, usize)> {
    let (mut s0, mut s1) = (0usize, 0usize);
    let rc = unsafe { ffi::vbx_shard_samples(lo, hi, frame_len, hop, &mut s0, &mut s1) };
    if rc != ffi::VBX_SUCCESS {
        return Err(GpuError { code: rc, message: last_error(ptr::null()) });
    }
    Ok((s0, s1))
}

/// `(offset, count, op)` per rank of the record gather as `rank` sees it (`vbx_gather_plan`; host arithmetic only).
pub fn gather_plan(rows: &[i64], rank: i32, dst: i32, row_doubles: usize) -> GpuResult<(Vec<i64>, Vec<i64>, Vec<i32>)> {
    let w = rows.len();
    let (mut off, mut cnt, mut op) = (vec![0i64; w], vec![0i64; w], vec![0i32; w]);
    let rc = unsafe { ffi::vbx_gather_plan(rows.as_ptr(), w as i32, rank, dst, row_doubles, off.as_mut_ptr(), cnt.as_mut_ptr(), op.as_mut_ptr()) };
    if rc != ffi::VBX_SUCCESS {
        return Err(GpuError { code: rc, message: last_error(ptr::null()) });
    }
    Ok((off, cnt, op))
}

/// This rank's RCCL communicator for the record gather (`vbx_comm_*`): one per process.  The 128-byte id comes from
/// [`Comm::unique_id`] on rank 0 and travels to the other ranks by whatever the application has (MPI, a TCP store, a file).
pub struct Comm<'g> {
    gpu: &'g Gpu,
    raw: *mut ffi::VbxComm,
    world: usize,
    rank: usize,
}

impl<'g> Comm<'g> {
    pub fn unique_id() -> GpuResult<[u8; ffi::VBX_UNIQUE_ID_BYTES]> {
        let mut id = [0u8; ffi::VBX_UNIQUE_ID_BYTES];
        let rc = unsafe { ffi::vbx_comm_unique_id(id.as_mut_ptr() as *mut c_void) };
        if rc != ffi::VBX_SUCCESS {
            return Err(GpuError { code: rc, message: last_error(ptr::null()) });
        }
        Ok(id)
    }

    /// Collective: every rank calls it with the same id.
    pub fn new(gpu: &'g Gpu, id: &[u8; ffi::VBX_UNIQUE_ID_BYTES], world: i32, rank: i32) -> GpuResult<Comm<'g>> {
        let mut raw: *mut ffi::VbxComm = ptr::null_mut();
        gpu.check(unsafe { ffi::vbx_comm_create(gpu.raw, id.as_ptr() as *const c_void, world, rank, &mut raw) })?;
        Ok(Comm { gpu, raw, world: world as usize, rank: rank as usize })
    }

    /// Live communicators of this process (`vbx_comm_live_count`): 1 while a `Comm` exists.
    pub fn live_count() -> i32 {
        unsafe { ffi::vbx_comm_live_count() }
    }

    /// Queues the gather of `records` (this rank's rows) to rank `dst` behind the kernels already queued on the context
    /// (`vbx_gather_records_f64`); `rows[r]` = frames of rank r; on `dst`, `out` receives every rank's rows in rank order
    /// (it may be the buffer `records` points into, at this rank's offset: no copy).  `slot` names the buffer for `wait`.
    ///
    /// `first_row`: rows of `records` before it are not sent (a shard's warm-up frames, `VbxShardPlan::warm`).  The sizes are
    /// checked here, on the host, against the gather's own transfer list (`vbx_gather_plan`): a short `out` or an inflated
    /// row count would otherwise let ncclRecv / the device copy write past the allocation from safe code.
    pub fn gather_records(&self, records: &Records, first_row: usize, rows: &[i64], dst: i32, out: Option<&DeviceBuf<f64>>, slot: i32) -> GpuResult<()> {
        assert_eq!(rows.len(), self.world, "one row count per rank");
        assert!(dst >= 0 && (dst as usize) < self.world, "dst must be a rank of the communicator");
        assert!(rows.iter().all(|&r| r >= 0), "row counts are not negative");
        assert!(first_row <= records.n_frames && rows[self.rank] as usize == records.n_frames - first_row,
                "rows[rank] must be the rows this rank sends");
        let (off, cnt, _) = gather_plan(rows, self.rank as i32, dst, records.record_ld)?;
        if self.rank == dst as usize {
            let need = (off[self.world - 1] + cnt[self.world - 1]) as usize;
            let o = out.expect("the destination rank passes the gathered buffer");
            assert!(o.len() >= need, "the gathered buffer holds {} doubles, the gather writes {}", o.len(), need);
        }
        let local = unsafe { records.data.as_ptr().add(first_row * records.record_ld) };
        self.gpu.check(unsafe {
            ffi::vbx_gather_records_f64(self.gpu.raw, self.raw, local, rows.as_ptr(), records.record_ld, dst,
                                        out.map_or(ptr::null_mut(), |o| o.as_mut_ptr()), slot)
        })
    }

    /// The formant tracks of `records` (the LAST `analyze` call on this `Gpu`: the plan's frames `[lo - warm, hi)`) continued
    /// from the previous rank's last row, and this rank's last row passed on (`vbx_comm_stitch_tracks_f64`): with it the
    /// gathered tracks of an utterance cut by the rank boundaries are the single-process scan's, bit for bit
    /// (src/spectrum.rs:357-369).  Queue [`Comm::gather_records`] after it with the same `slot`.
    pub fn stitch_tracks(&self, records: &Records, params: &AnalysisParams, plan: &ffi::VbxShardPlan, slot: i32) -> GpuResult<()> {
        assert!(params.formant_order > 0, "the records carry no formant tracks");
        assert_eq!(records.n_frames, plan.hi - plan.lo + plan.warm, "the records are the plan's frames [lo - warm, hi)");
        assert!(records.record_ld >= 2 + 2 * params.est_init.len());
        let formants = unsafe { records.data.as_mut_ptr().add(2) } as *mut ffi::VbxResonance;      // record columns: pitch, then the formants
        self.gpu.check(unsafe {
            ffi::vbx_comm_stitch_tracks_f64(self.gpu.raw, self.raw, formants, records.n_frames, records.record_ld, plan, ptr::null_mut(), slot)
        })
    }

    /// The context's stream waits (on the device) for the gather that used `slot`: call before overwriting that buffer.
    pub fn wait(&self, slot: i32) -> GpuResult<()> {
        self.gpu.check(unsafe { ffi::vbx_comm_wait(self.gpu.raw, self.raw, slot) })
    }

    /// The host waits for every queued gather.
    pub fn sync(&self) -> GpuResult<()> {
        self.gpu.check(unsafe { ffi::vbx_comm_sync(self.raw) })
    }

    /// Loopback self-test of the RCCL path on this GPU (`vbx_comm_selftest`).
    pub fn selftest(&self, n_doubles: usize) -> GpuResult<()> {
        self.gpu.check(unsafe { ffi::vbx_comm_selftest(self.gpu.raw, self.raw, n_doubles) })
    }
}

impl<'g> Drop for Comm<'g> {
    fn drop(&mut self) {
        unsafe { ffi::vbx_comm_destroy(self.raw) };
    }
}
