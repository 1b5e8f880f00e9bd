//! vox_box_hip -- vox_box 0.3.0's per-frame DSP traits over libvoxbox_hip.so (MI355X / gfx950).
//!
//! The reference crate exposes its hot path as extension traits on slices that user code calls once per
//! frame (`examples/pitch_detection.rs:23-30`, `tests/lib.rs:71-83`).  This crate keeps those traits --
//! `Autocorrelate` (src/periodic.rs:265-274), `LPC` (src/spectrum.rs:50-55), `Pitched` (src/periodic.rs:356-358),
//! `ToResonance` (src/spectrum.rs:195-210), `MFCC` (src/spectrum.rs:371-373) -- and `find_formants`
//! (src/lib.rs:40), and moves the receiver from one `&[f64]` to a [`gpu::FrameBatch`]: all frames of a recording
//! resident in HBM.  Two ways to call:
//!
//! * **batched** (`FrameBatch::autocorrelate_all`, `pitch_all`, `lpc_praat_all`, `mfcc_all`, `find_formants`):
//!   one library call = the user's whole frame loop;
//! * **fused** (`FrameBatch::analyze` / `PcmBatch::analyze`): pitch + LPC + find_formants + MFCC of every frame as one
//!   record per frame from one call (`vbx_analyze_frames_f64` / `_pcm16`), the buffer [`gpu::Comm::gather_records`] sends
//!   to rank 0 when the recording is sharded over the GPUs of a node ([`gpu::shard_range`], [`gpu::shard_samples`]);
//! * **tracker** (`ResonanceRows::estimate_formants_all`, `ResonanceRows::formant_extractor`): `EstimateFormants` /
//!   `FormantExtractor` (src/spectrum.rs:216-369) over resonance rows on the device;
//! * **drop-in** (`FrameBatch::frames()` yields [`gpu::GpuFrame`] views that implement the crate's traits):
//!   the user's loop stays as written, the first call of a method computes the whole batch on the GPU and the
//!   per-frame calls read their row of the cached result.
//!
//! There is no CPU fallback: `Gpu::new` fails without a gfx950 device.
//!
//! This crate is shipped as source (the build image has no Rust toolchain); `ffi.rs` is generated from
//! `include/voxbox_hip.h` by `tools/gen_rust_ffi.py`.

pub mod ffi;
pub mod gpu;

pub use gpu::{find_formants, gather_plan, shard_range, shard_samples, AnalysisParams, Comm, DeviceBuf, FormantExtractor, FrameBatch, FrameStatus,
              Frames, Gpu, GpuError, GpuEstimates, GpuFrame, PcmBatch, PolyBatch, Records, ResonanceRows, RootRow, RootRows};
