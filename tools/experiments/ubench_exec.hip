// Does the vector ALU skip 16-lane passes whose EXEC bits are all zero?  FP64 FMA / compare-select chains with 64, 32,
// 16 and 1 active lanes (exec narrowed by a real branch).  If it does, wave-uniform scalar work (Brent's arithmetic in the
// pitch refinement) could run under a narrowed EXEC at a fraction of its issue cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int ACTIVE, int DEP> __global__ void k(double *out, double seed) {
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    if ((int)(threadIdx.x & 63) < ACTIVE) {
        for (int i = 0; i < ITER; i++) {
            if (DEP) { a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); }
            else { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int ACTIVE, int DEP> void run(int waves_per_simd) {
    double *d; hipMalloc(&d, 1 << 24);
    int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<ACTIVE, DEP><<<blocks, 64>>>(d, 1.5); hipDeviceSynchronize();
    hipEventRecord(a); k<ACTIVE, DEP><<<blocks, 64>>>(d, 1.5); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)waves_per_simd * ITER * 8;
    printf("active %2d %s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", ACTIVE, DEP ? "dependent  " : "independent",
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<64, 0>(w); run<32, 0>(w); run<16, 0>(w); run<1, 0>(w);
        run<64, 1>(w); run<16, 1>(w); run<1, 1>(w);
    }
    return 0;
}
