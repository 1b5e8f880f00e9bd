// v_mfma_f64_16x16x4_f64: layout check + issue rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void layout(const double *A, const double *B, double *C) {   // A[16][4] row-major, B[4][16], C[16][16]
    const int lane = threadIdx.x;
    // guide: A/B as the f32 16x16x4 form, one f64 per lane: A: row = lane&15, k = lane>>4; B: col = lane&15, k = lane>>4
    const double a = A[(lane & 15) * 4 + (lane >> 4)];
    const double b = B[(lane >> 4) * 16 + (lane & 15)];
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    // C/D: col = lane&15, row = (lane>>4) + 4*reg
    for (int r = 0; r < 4; r++) C[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}
template <int NACC>
__global__ void rate(double *out, int iters) {
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    d4 acc[NACC];
    for (int j = 0; j < NACC; j++) acc[j] = d4{0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    double s = 0;
    for (int j = 0; j < NACC; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 64 + lane] = s;
}
int main() {
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 64; i++) { A[i] = sin(i + 1.0); B[i] = cos(2.0 * i + 0.5); }
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) for (int k = 0; k < 4; k++) R[m * 16 + n] += A[m * 4 + k] * B[k * 16 + n];
    double *dA, *dB, *dC; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(dA, dB, dC); hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; i++) err = fmax(err, fabs(C[i] - R[i]));
    printf("layout max err %.3e\n", err);
    double *d; hipMalloc(&d, 1 << 24);
    for (int w : {1, 2, 4}) {
        const int blocks = 256 * 4 * w, iters = 20000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        rate<4><<<blocks, 64>>>(d, iters); hipDeviceSynchronize();
        hipEventRecord(a); rate<4><<<blocks, 64>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double n = (double)w * iters * 4;
        printf("waves/SIMD %d NACC 4: %.3f ms -> %.1f cycles per MFMA per SIMD at 2.4 GHz; %.1f TFLOP/s\n", w, ms, ms * 1e-3 * 2.4e9 / n,
               (double)blocks * iters * 4 * 2048 / (ms * 1e-3) / 1e12);
        hipEventRecord(a); rate<1><<<blocks, 64>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("waves/SIMD %d NACC 1 (dependent): %.1f cycles per MFMA per SIMD\n", w, ms * 1e-3 * 2.4e9 / ((double)w * iters));
    }
    return 0;
}
