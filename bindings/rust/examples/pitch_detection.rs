//! The reference's examples/pitch_detection.rs (150 Hz sine at 44.1 kHz, Hanning Windower 2048 / 1024, `pitch`
//! per frame) with the frames on the GPU.  Expected output: frame 0 -> 149.99998 Hz, 0.99975.
//!
//!   VOXBOX_HIP_LIB_DIR=../../vox_box.rs_amd/lib cargo run --release --example pitch_detection
extern crate sample;
extern crate vox_box;
extern crate vox_box_hip;

use sample::{Signal, ToSampleSlice};
use vox_box::periodic::{Hanning, Pitch, Pitched};
use vox_box::waves::MaxAmplitude;
use vox_box_hip::{FrameBatch, Gpu};

fn main() {
    let signal = sample::signal::rate(44100.).const_hz(150.0).sine();
    let vector: Vec<[f64; 1]> = signal.take(2048 + 1).collect();
    let samples: &[f64] = vector.to_sample_slice();
    let maxima: f64 = samples.max_amplitude();

    let gpu = Gpu::new(0).expect("no MI355X: the library has no CPU path");
    // window::Windower::hanning(&vector[..], 2048, 1024), resident in HBM
    let batch = FrameBatch::windower_hanning(&gpu, samples, 2048, 1024).unwrap();

    // drop-in: the reference's loop body, one frame at a time (the first call analyses every frame)
    let mut pitches_out: Vec<Vec<Pitch<f64>>> = Vec::new();
    for frame in batch.frames() {
        pitches_out.push(frame.pitch::<Hanning>(44100., 0.2, maxima, maxima, 100., 500.));
    }
    println!("pitches_out: {:?}", pitches_out);

    // batched: the PitchExtractor output (src/periodic.rs:337-353) of every frame in one call
    let (top, count, status) = batch.pitch_all(44100., 0.2, 100., 500., 1).unwrap();
    println!("top candidate per frame: {:?} (candidates: {:?}, status: {:?})", top, count, status);
}
