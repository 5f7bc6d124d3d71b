// Issue rate / sustained FLOP rate of the fp16 MFMA shapes on gfx950 with every CU busy: one dependent chain vs
// independent accumulators, 1 or 2 waves per SIMD, trivial vs random operands (the chip holds its clock down under
// MFMA load, by an amount that depends on the data and on the shape -- MI355X_MICROARCH.md 'DVFS give-back').
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int SHAPE, int NACC, int RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    h8 a[4], b[4];
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = hashu(threadIdx.x * 64 + s * 8 + i + blockIdx.x * 7919);
            a[s][i] = RANDOM ? (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f)) : (_Float16)1.0f;
            b[s][i] = RANDOM ? (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.f)) : (_Float16)1.0f;
        }
    f16v c32[NACC];
    f4v c16[NACC * 4];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) c32[n][i] = 0.f;
    for (int n = 0; n < NACC * 4; ++n) for (int i = 0; i < 4; ++i) c16[n][i] = 0.f;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if (SHAPE == 32) c32[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u + n) & 3], c32[n], 0, 0, 0);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)     // 4 x (16x16x32) = the flops of TWO 32x32x16
                        c16[4 * n + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q], b[n & 3], c16[4 * n + q], 0, 0, 0);
                }
            }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += c32[n][i];
    for (int n = 0; n < NACC * 4; ++n) for (int i = 0; i < 4; ++i) s += c16[n][i];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 17) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
template <int SHAPE, int NACC, int RANDOM>
void run(int wgs_per_cu, float* d, unsigned long long* dc) {
    const int iters = 16384 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<SHAPE, NACC, RANDOM><<<256 * wgs_per_cu, 256>>>(d, iters, dc);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    k<SHAPE, NACC, RANDOM><<<256 * wgs_per_cu, 256>>>(d, iters, dc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8 * NACC * (SHAPE == 32 ? 1 : 2);      // in units of one 32x32x16 worth of flops
    const double tf = 2.0 * 32 * 32 * 16 * nm * 4 * 256 * wgs_per_cu / (ms * 1e-3) / 1e12;
    printf("%dx%d acc=%d waves/SIMD=%d %-7s: %5.1f cycles per 32x32x16-equivalent per wave, %4.0f TFLOP/s, in-kernel clock %.2f GHz\n",
           SHAPE, SHAPE, NACC, wgs_per_cu, RANDOM ? "random" : "ones", c[0] / nm, tf, c[0] / (c[1] * 10.0));
}
int main() {
    float* d; unsigned long long* dc;
    hipMalloc(&d, 4096); hipMalloc(&dc, 16);
    run<32, 1, 0>(1, d, dc); run<32, 4, 0>(1, d, dc); run<32, 4, 1>(1, d, dc); run<32, 4, 1>(2, d, dc);
    run<16, 4, 0>(1, d, dc); run<16, 4, 1>(1, d, dc); run<16, 4, 1>(2, d, dc);
    return 0;
}
