// HBM read bandwidth of two access patterns over the same 4 GB (gfx950): (A) config 2's -- every lane owns 64 contiguous
// bytes of a 4 KB frame and fetches them with four 16-byte loads (each load instruction touches 64 cache lines, every line is
// touched by four instructions); (B) fully coalesced -- lane i reads bytes [16 i, 16 i + 16) of consecutive 1 KB blocks.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/ubench_bw.hip -o tools/experiments/ubench_bw
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int FPW = 16;   // frames per wavefront, as config 2

__global__ __launch_bounds__(64) void pat_a(const double2 *__restrict__ x, long n_frames, double *out) {
    const int lane = threadIdx.x;
    const long f0 = (long)blockIdx.x * FPW;
    double acc = 0.0;
    for (int g = 0; g < FPW && f0 + g < n_frames; g++) {
        const double2 *p = x + (f0 + g) * 256 + lane * 4;          // 4 KB frame = 256 double2; lane owns 4 of them
        double2 v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = p[e];
#pragma unroll
        for (int e = 0; e < 4; e++) acc += v[e].x * v[e].y;
    }
    if (acc == 12345.678) out[0] = acc;
}

__global__ __launch_bounds__(64) void pat_b(const double2 *__restrict__ x, long n_frames, double *out) {
    const int lane = threadIdx.x;
    const long f0 = (long)blockIdx.x * FPW;
    double acc = 0.0;
    for (int g = 0; g < FPW && f0 + g < n_frames; g++) {
        const double2 *p = x + (f0 + g) * 256 + lane;              // block e: lanes read consecutive 16 bytes
        double2 v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = p[64 * e];
#pragma unroll
        for (int e = 0; e < 4; e++) acc += v[e].x * v[e].y;
    }
    if (acc == 12345.678) out[0] = acc;
}

int main() {
    const long F = 1000000;
    double2 *x; double *out;
    hipMalloc(&x, F * 4096); hipMalloc(&out, 8);
    hipMemset(x, 0, F * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int pat = 0; pat < 2; pat++) {
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(a);
            if (pat == 0) hipLaunchKernelGGL(pat_a, dim3((F + FPW - 1) / FPW), dim3(64), 0, 0, x, F, out);
            else hipLaunchKernelGGL(pat_b, dim3((F + FPW - 1) / FPW), dim3(64), 0, 0, x, F, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("pattern %c: %.3f ms  %.2f TB/s\n", pat ? 'B' : 'A', best, F * 4096.0 / best / 1e9);
    }
    return 0;
}
