// Issue rate of v_mfma_f32_32x32x16_f16 on gfx950: one dependent chain vs independent accumulators, 1 or 2 waves
// per SIMD, every CU busy (so the clock is what a real kernel sees).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    f16v c[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int wgs_per_cu, float* d, unsigned long long* dc) {
    const int iters = 4096 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256 * wgs_per_cu, 256>>>(d, iters, dc);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    k<NACC><<<256 * wgs_per_cu, 256>>>(d, iters, dc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8 * NACC;
    const double tf = 2.0 * 32 * 32 * 16 * nm * 4 * 256 * wgs_per_cu / (ms * 1e-3) / 1e12;
    printf("acc=%d waves/SIMD=%d: %.1f cycles per MFMA per wave, %.0f TFLOP/s, wall %.3f ms, clock ~%.2f GHz\n", NACC, wgs_per_cu,
           c / nm, tf, ms, c / (ms * 1e6));
}
int main() {
    float* d; unsigned long long* dc;
    hipMalloc(&d, 4096); hipMalloc(&dc, 8);
    run<1>(1, d, dc); run<4>(1, d, dc); run<1>(2, d, dc); run<4>(2, d, dc);
    return 0;
}
