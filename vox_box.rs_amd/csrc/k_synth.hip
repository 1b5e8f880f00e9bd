// k_synth.hip -- deterministic speech-like synthetic audio, closed form per sample
// (bench / test utility; DESIGN.md "synthetic signal").  Every sample depends only on its
// absolute index and the seed, so any rank can generate its own shard in place.
//
//   t      = s / sr;  tau = t mod 2 s
//   f0(t)  : 90 -> 250 -> 90 Hz triangular glide, 2 s period (340 cycles per period)
//   voiced : 0.25 * sum_{h=1..30} (1/h) * G(h*f0) * sin(2 pi h Phi(t))
//            G(f) = sum_k 1/sqrt(1 + ((f - F_k)/(B_k/2))^2),  F = 700,1220,2600,3300  B = 130,70,160,250
//   noise  : splitmix64(seed + s * golden) -> uniform(-1, 1)
//   every 5th second (floor(t) mod 5 == 4) is unvoiced: 0.05 * noise; otherwise voiced + 0.0025 * noise
#include "vbx_device.hpp"
#include "vbx_kernels.hpp"

namespace vbx {

__device__ __forceinline__ double synth_sample(uint64_t s, double sr, uint64_t seed) {
    uint64_t z = seed + s * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const double noise = 2.0 * ((double)(z >> 11) * (1.0 / 9007199254740992.0)) - 1.0;

    const double t = (double)s / sr;
    const double sec = floor(t);
    const long isec = (long)sec;
    if (isec % 5 == 4) return 0.05 * noise;
    const double tau = t - 2.0 * floor(t * 0.5);
    double f0, phi;
    if (tau < 1.0) { f0 = 90.0 + 160.0 * tau; phi = 90.0 * tau + 80.0 * tau * tau; }
    else { const double u = tau - 1.0; f0 = 250.0 - 160.0 * u; phi = 170.0 + 250.0 * u - 80.0 * u * u; }
    const double th = 2.0 * M_PI * (phi - floor(phi));
    double sn, cs;
    sincos(th, &sn, &cs);
    const double two_c = 2.0 * cs;
    double s_prev = 0.0, s_cur = sn;       // sin(0*th), sin(1*th)
    double acc = 0.0;
    const double F[4] = {700.0, 1220.0, 2600.0, 3300.0};
    const double HB[4] = {65.0, 35.0, 80.0, 125.0};
#pragma unroll 1
    for (int h = 1; h <= 30; h++) {
        const double fh = (double)h * f0;
        double g = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const double d = (fh - F[k]) / HB[k]; g += rsqrt(1.0 + d * d); }
        acc += (g / (double)h) * s_cur;
        const double s_next = two_c * s_cur - s_prev;
        s_prev = s_cur; s_cur = s_next;
    }
    return 0.25 * acc + 0.0025 * noise;
}

__global__ void synth_kernel(double *__restrict__ out, size_t n_samples, uint64_t sample_offset, double sr, uint64_t seed) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < n_samples; i += step) out[i] = synth_sample(sample_offset + i, sr, seed);
}

void launch_synth(hipStream_t s, double *out, size_t n_samples, uint64_t sample_offset, double sample_rate, uint64_t seed) {
    const int bs = 256;
    size_t blocks = (n_samples + bs - 1) / bs;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(bs), 0, s, out, n_samples, sample_offset, sample_rate, seed);
}

}  // namespace vbx
