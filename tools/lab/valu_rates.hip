// Lab: issue rate of the vector opcodes the bf16 arithmetics could use (dot2 forms, packed f16, perm, f16 transcendentals)
// against v_fma_f32 / v_pk_fma_f32, at 1 / 2 / 4 waves per SIMD, no memory traffic.  Build: hipcc -O3 --offload-arch=gfx950
// tools/lab/valu_rates.hip -o /tmp/valu_rates.  Output: ns and shader cycles (s_memtime) per wave-instruction and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define OP3(name, text)                                                                                  \
    struct name { static __device__ __forceinline__ void run(unsigned& d, unsigned a, unsigned b) {      \
        asm volatile(text : "+v"(d) : "v"(a), "v"(b)); } static const char* nm() { return #name; } };
OP3(fma_f32, "v_fma_f32 %0, %1, %2, %0")
OP3(dot2_f32_bf16, "v_dot2_f32_bf16 %0, %1, %2, %0")
OP3(dot2c_f32_bf16, "v_dot2c_f32_bf16 %0, %1, %2")
OP3(dot2_f32_f16, "v_dot2_f32_f16 %0, %1, %2, %0")
OP3(pk_fma_f16, "v_pk_fma_f16 %0, %1, %2, %0")
OP3(pk_mul_f16, "v_pk_mul_f16 %0, %1, %2")
OP3(pk_max_f16, "v_pk_max_f16 %0, %1, %2")
OP3(perm_b32, "v_perm_b32 %0, %1, %2, %0")
OP3(cvt_pk_f16_f32, "v_cvt_pk_f16_f32 %0, %1, %2")
OP3(cvt_pk_bf16_f32, "v_cvt_pk_bf16_f32 %0, %1, %2")
OP3(exp_f32, "v_exp_f32 %0, %1")
OP3(exp_f16, "v_exp_f16 %0, %1")
OP3(exp_f16_sdwa, "v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1")
OP3(fma_mix_f32, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]")
OP3(fma_mixlo_f16, "v_fma_mixlo_f16 %0, %1, %2, %0")
OP3(lshl_or, "v_lshl_or_b32 %0, %1, 16, %2")
OP3(and_or, "v_and_or_b32 %0, %1, %2, %0")
OP3(bfi, "v_bfi_b32 %0, %1, %2, %0")

struct pk_fma_f32 { static __device__ __forceinline__ void run2(unsigned long long& d, unsigned long long a, unsigned long long b) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b)); } static const char* nm() { return "pk_fma_f32"; } };

template <class OP>
__global__ __launch_bounds__(256) void k(unsigned* out, long long* cyc, int iters, unsigned s) {
    unsigned d[8], a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = 0x3c003c00u + threadIdx.x + i; a[i] = 0x38003800u + s * i; b[i] = 0x3a003a00u + (s ^ i); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) OP::run(d[i], a[(i + u) & 7], b[(i + 3 * u) & 7]);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    unsigned t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t ^= d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(256) void k2(unsigned* out, long long* cyc, int iters, unsigned s) {
    unsigned long long d[8], a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = 0x3f8000003f800000ull + threadIdx.x + i; a[i] = 0x3f0000003f000000ull + s * i; b[i] = 0x3f4000003f400000ull + (s ^ i); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) pk_fma_f32::run2(d[i], a[(i + u) & 7], b[(i + 3 * u) & 7]);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t ^= d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)t;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static unsigned* g_out; static long long* g_cyc;
template <class F>
static void bench(const char* name, F launch) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        dim3 grid(256 * wps), blk(256);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            launch(grid, blk, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        std::vector<long long> c(256 * wps * 4);
        hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        const double per_wave_instr = (double)c[c.size() / 2] / ((double)iters * 64);
        printf("%-16s waves/SIMD=%d  %8.3f ms  %6.2f ns per instr and SIMD  %6.2f cycles per instr of one wave  %6.2f cycles per instr and SIMD\n",
               name, wps, best, best * 1e6 / ((double)iters * 64 * wps), per_wave_instr, per_wave_instr / wps);
    }
}
#define BENCH(OP) bench(OP::nm(), [](dim3 g, dim3 b, int it) { k<OP><<<g, b>>>(g_out, g_cyc, it, 3u); });
int main() {
    hipMalloc(&g_out, 256 * 4 * 256 * 4); hipMalloc(&g_cyc, 256 * 4 * 4 * 8);
    BENCH(fma_f32)
    bench("pk_fma_f32", [](dim3 g, dim3 b, int it) { k2<<<g, b>>>(g_out, g_cyc, it, 3u); });
    BENCH(dot2_f32_bf16) BENCH(dot2c_f32_bf16) BENCH(dot2_f32_f16) BENCH(pk_fma_f16) BENCH(pk_mul_f16) BENCH(pk_max_f16)
    BENCH(perm_b32) BENCH(cvt_pk_f16_f32) BENCH(cvt_pk_bf16_f32) BENCH(exp_f32) BENCH(exp_f16) BENCH(exp_f16_sdwa)
    BENCH(fma_mix_f32) BENCH(fma_mixlo_f16) BENCH(lshl_or) BENCH(and_or) BENCH(bfi)
    return 0;
}
