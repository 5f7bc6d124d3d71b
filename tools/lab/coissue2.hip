// Second co-issue microbenchmark (round 3): what do fillers cost in the MFMA gaps of one wave per SIMD when the setting is closer
// to the product kernels -- accumulators in AGPRs, the 16x16x32 shape (pairs), scalar-operand VALU forms, v_accvgpr_read fillers,
// rotating operand registers?  Every loop body is ONE asm volatile block (the order in the binary is the order written).
//   hipcc -O3 --offload-arch=gfx950 tools/lab/coissue2.hip -o build/labs/coissue2 && build/labs/coissue2
// Prints shader cycles (s_memtime) per 32 matrix cycles ("slot": one 32x32x16 MFMA or two 16x16x32), median over workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define SLOTS 16
#define STR(x) #x
#define XSTR(x) STR(x)
// one slot = MFMA part + filler part; 16 slots per iteration, accumulators alternate (g & 1)
#define M32(acc) "v_mfma_f32_32x32x16_f16 %[" #acc "], %[a0], %[b0], %[" #acc "]\n"
#define M32R(acc, ai, bi) "v_mfma_f32_32x32x16_f16 %[" #acc "], %[a" #ai "], %[b" #bi "], %[" #acc "]\n"
#define M16(acc) "v_mfma_f32_16x16x32_f16 %[" #acc "], %[a0], %[b0], %[" #acc "]\n"
#define F(r) "v_fma_f32 %[f" #r "], %[f" #r "], %[c1], %[c2]\n"
#define FS(r) "v_fma_f32 %[f" #r "], %[f" #r "], %[s1], %[c2]\n"
#define AR(r, acc) "v_accvgpr_read_b32 %[f" #r "], %[g" #r "]\n"

template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters, float sc) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x;
    f32x16 q0, q1; f32x4 r0, r1, r2, r3;
    for (int i = 0; i < 16; ++i) { q0[i] = in[(tid + i) & 4095]; q1[i] = in[(tid + 16 + i) & 4095]; }
    for (int i = 0; i < 4; ++i) { r0[i] = in[(tid + i) & 4095]; r1[i] = in[(tid + 4 + i) & 4095]; r2[i] = in[(tid + 8 + i) & 4095]; r3[i] = in[(tid + 12 + i) & 4095]; }
    h8 a0, a1, a2, a3, b0, b1, b2, b3;
    for (int i = 0; i < 8; ++i) {
        a0[i] = (_Float16)in[(tid * 8 + i) & 4095]; a1[i] = (_Float16)in[(tid * 8 + i + 100) & 4095]; a2[i] = (_Float16)in[(tid * 8 + i + 200) & 4095]; a3[i] = (_Float16)in[(tid * 8 + i + 300) & 4095];
        b0[i] = (_Float16)in[(tid * 8 + i + 2048) & 4095]; b1[i] = (_Float16)in[(tid * 8 + i + 2148) & 4095]; b2[i] = (_Float16)in[(tid * 8 + i + 2248) & 4095]; b3[i] = (_Float16)in[(tid * 8 + i + 2348) & 4095];
    }
    float g0 = in[(tid + 7) & 4095], g1 = in[(tid + 71) & 4095], g2 = in[(tid + 135) & 4095], g3 = in[(tid + 199) & 4095];
    float f0 = in[tid & 4095], f1 = in[(tid + 64) & 4095], f2 = in[(tid + 128) & 4095], f3 = in[(tid + 192) & 4095];
    const float c1 = 0.9999f, c2 = 1e-4f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#define OPS_V : [q0] "+v"(q0), [q1] "+v"(q1), [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3) \
              : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [c1] "v"(c1), [c2] "v"(c2), [s1] "s"(sc)
#define OPS_A : [q0] "+a"(q0), [q1] "+a"(q1), [r0] "+a"(r0), [r1] "+a"(r1), [r2] "+a"(r2), [r3] "+a"(r3), [g0] "+a"(g0), [g1] "+a"(g1), [g2] "+a"(g2), [g3] "+a"(g3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3) \
              : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [c1] "v"(c1), [c2] "v"(c2), [s1] "s"(sc)
#define FOUR F(0) F(1) F(2) F(3)
#define REP8(x) x x x x x x x x
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) asm volatile(REP8(M32(q0) M32(q1)) OPS_V);                                              // bare 32x32x16, VGPR accumulators
        else if constexpr (V == 1) asm volatile(REP8(M32(q0) FOUR M32(q1) FOUR) OPS_V);                               // + 4 fma per slot
        else if constexpr (V == 2) asm volatile(REP8(M32(q0) FOUR M32(q1) FOUR) OPS_A);                               // accumulators in AGPRs
        else if constexpr (V == 3) asm volatile(REP8(M16(r0) M16(r1) M16(r2) M16(r3)) OPS_V);                         // bare 16x16x32 (2 per slot)
        else if constexpr (V == 4) asm volatile(REP8(M16(r0) M16(r1) FOUR M16(r2) M16(r3) FOUR) OPS_V);               // pair, then 4 fma
        else if constexpr (V == 5) asm volatile(REP8(M16(r0) F(0) F(1) M16(r1) F(2) F(3) M16(r2) F(0) F(1) M16(r3) F(2) F(3)) OPS_V);   // 2 fma behind each
        else if constexpr (V == 6) asm volatile(REP8(M16(r0) M16(r1) FOUR M16(r2) M16(r3) FOUR) OPS_A);               // pair + 4, AGPR accumulators
        else if constexpr (V == 7) asm volatile(REP8(M32(q0) FS(0) FS(1) FS(2) FS(3) M32(q1) FS(0) FS(1) FS(2) FS(3)) OPS_V);           // fma with an SGPR operand
        else if constexpr (V == 8) asm volatile(REP8(M32(q0) AR(0, r0) AR(1, r1) AR(2, r2) AR(3, r3) M32(q1) AR(0, r0) AR(1, r1) AR(2, r2) AR(3, r3)) OPS_A);   // v_accvgpr_read fillers
        else if constexpr (V == 9) asm volatile(REP8(M32R(q0, 0, 0) FOUR M32R(q1, 1, 1) FOUR) REP8(M32R(q0, 2, 2) FOUR M32R(q1, 3, 3) FOUR) OPS_V);             // rotating operands (32 slots)
        else if constexpr (V == 10) asm volatile(REP8(M16(r0) M16(r1) F(0) F(1) F(2) M16(r2) M16(r3) F(0) F(1) F(2)) OPS_V);            // pair + 3
        else if constexpr (V == 11) asm volatile(REP8(M16(r0) M16(r1) F(0) F(1) M16(r2) M16(r3) F(0) F(1)) OPS_V);                      // pair + 2
        else if constexpr (V == 12) asm volatile(REP8(M16(r0) M16(r1) FOUR F(0) F(1) M16(r2) M16(r3) FOUR F(0) F(1)) OPS_V);            // pair + 6
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = f0 + f1 + f2 + f3 + g0 + g1 + g2 + g3;
    for (int i = 0; i < 16; ++i) s += q0[i] + q1[i];
    for (int i = 0; i < 4; ++i) s += r0[i] + r1[i] + r2[i] + r3[i];
    if (s == 12345.678f) out[tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V>
void run(const char* name, int slots, const float* in, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 0; w < 2; ++w) k<V><<<256, 256, 160 * 1024>>>(in, out, cyc, iters, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s %7.2f cycles per slot of 32 matrix cycles\n", name, (double)h[128] / ((double)iters * slots));
}
int main() {
    float *in, *out; unsigned long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u >> 20) & 0xfff) * 1e-3f - 2.f;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0>("32x32x16 bare, VGPR acc", 16, in, out, cyc);
    run<1>("32x32x16 + 4 fma / slot, VGPR acc", 16, in, out, cyc);
    run<2>("32x32x16 + 4 fma / slot, AGPR acc", 16, in, out, cyc);
    run<3>("16x16x32 bare (2 per slot)", 16, in, out, cyc);
    run<4>("16x16x32 pair, then 4 fma", 16, in, out, cyc);
    run<5>("16x16x32, 2 fma behind each", 16, in, out, cyc);
    run<6>("16x16x32 pair + 4 fma, AGPR acc", 16, in, out, cyc);
    run<7>("32x32x16 + 4 fma with SGPR operand", 16, in, out, cyc);
    run<8>("32x32x16 + 4 v_accvgpr_read", 16, in, out, cyc);
    run<9>("32x32x16 + 4 fma, rotating A/B registers", 32, in, out, cyc);
    run<10>("16x16x32 pair + 3 fma", 16, in, out, cyc);
    run<11>("16x16x32 pair + 2 fma", 16, in, out, cyc);
    run<12>("16x16x32 pair + 6 fma", 16, in, out, cyc);
    return 0;
}
