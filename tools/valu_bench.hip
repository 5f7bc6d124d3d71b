// Microbenchmark: fp32 FMA issue rate (v_fma_f32 vs v_pk_fma_f32) at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
// DIST: number of independent accumulators cycled through (1 = every FMA depends on the previous one)
template <int PK, int DIST = 8>
__global__ void k(float* out, int iters, float s) {
    f32x2 a[8], b[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = f32x2{(float)threadIdx.x + i, (float)i}; b[i] = f32x2{s * i, s * 0.5f}; c[i] = f32x2{1.0001f + i * 1e-6f, 0.9999f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i % DIST]) : "v"(b[(i + u) & 7]), "v"(c[(i + 3 * u) & 7]));
                else { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][0]) : "v"(b[(i + u) & 7][0]), "v"(c[(i + 3 * u) & 7][0]));
                       asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][1]) : "v"(b[(i + u) & 7][1]), "v"(c[(i + 3 * u) & 7][1])); }
            }
        }
    }
    float t = 0; for (int i = 0; i < 8; ++i) t += a[i][0] + a[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wps : {1, 2, 4, 8}) {
            dim3 grid(256 * wps), blk(256);     // wps blocks of 4 waves per CU -> wps waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (pk) k<1><<<grid, blk>>>(out, iters, 1e-3f); else k<0><<<grid, blk>>>(out, iters, 1e-3f);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fma_lanes = (double)iters * 64 * 2 * 64 * 4 * wps * 256;   // per-lane fp32 FMAs
            printf("%s waves/SIMD=%d  %.2f ms  %.1f TFLOP/s  (%.2f cycles per wave64 %s at 2.4 GHz)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms,
                   2 * fma_lanes / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * (pk ? 1 : 2) * wps), pk ? "pk-instr" : "instr");
        }
    for (int dist : {1, 2, 3, 4})
        for (int wps : {1, 2}) {
            dim3 grid(256 * wps), blk(256);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (dist == 1) k<1, 1><<<grid, blk>>>(out, iters, 1e-3f);
                else if (dist == 2) k<1, 2><<<grid, blk>>>(out, iters, 1e-3f);
                else if (dist == 3) k<1, 3><<<grid, blk>>>(out, iters, 1e-3f);
                else k<1, 4><<<grid, blk>>>(out, iters, 1e-3f);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("v_pk_fma_f32 dependency distance %d, waves/SIMD=%d: %.2f ms (%.2f cycles per pk-instr per SIMD at 2.4 GHz)\n", dist, wps, ms,
                   ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
        }
    return 0;
}
