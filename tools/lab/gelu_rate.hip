// Micro-benchmark: how fast does the GELU + split (split_math.h, gelu_micro2: 240 vector instructions per 32 x 32 tile of a
// wave) issue on its own -- no MFMA, no LDS -- at one and at two waves per SIMD, in the micro-step order the fused kernels use
// (two register pairs alternating) and with all eight pairs interleaved step by step?
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -I audioset-convnext-inf_amd/csrc tools/lab/gelu_rate.hip -o build/labs/gelu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "split_math.h"
using namespace acx;

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ in, unsigned* __restrict__ out, unsigned long long* cyc, int iters, float sinv, float kh) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x;
    f32x16 Xv;
    for (int i = 0; i < 16; ++i) Xv[i] = in[(tid * 16 + i) & 4095] / sinv;
    const GeluK2 gk = gelu_k2(sinv, kh);
    unsigned uh[8], ul[8], acc = 0;
    GeluState2 gs[8];
    const unsigned long long t0 = stamp();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // the fused kernels' order: pairs (2g, 2g+1) alternate, step by step
#pragma unroll
            for (int mm = 0; mm < 64; ++mm) {
                const int pr = 2 * (mm / 16) + (mm & 1), st = (mm % 16) >> 1;
                GeluState2& g = gs[mm & 1];
                if (st == 0) { g.ax = Xv[2 * pr]; g.ay = Xv[2 * pr + 1]; gelu_micro2<0>(g, gk, uh[pr], ul[pr]); }
                else if (st == 1) gelu_micro2<1>(g, gk, uh[pr], ul[pr]);
                else if (st == 2) gelu_micro2<2>(g, gk, uh[pr], ul[pr]);
                else if (st == 3) gelu_micro2<3>(g, gk, uh[pr], ul[pr]);
                else if (st == 4) gelu_micro2<4>(g, gk, uh[pr], ul[pr]);
                else if (st == 5) gelu_micro2<5>(g, gk, uh[pr], ul[pr]);
                else if (st == 6) gelu_micro2<6>(g, gk, uh[pr], ul[pr]);
                else gelu_micro2<7>(g, gk, uh[pr], ul[pr]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {                    // all eight pairs abreast: step s of every pair, then step s + 1
#pragma unroll
            for (int st = 0; st < 8; ++st) {
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {
                    GeluState2& g = gs[pr];
                    if (st == 0) { g.ax = Xv[2 * pr]; g.ay = Xv[2 * pr + 1]; gelu_micro2<0>(g, gk, uh[pr], ul[pr]); }
                    else if (st == 1) gelu_micro2<1>(g, gk, uh[pr], ul[pr]);
                    else if (st == 2) gelu_micro2<2>(g, gk, uh[pr], ul[pr]);
                    else if (st == 3) gelu_micro2<3>(g, gk, uh[pr], ul[pr]);
                    else if (st == 4) gelu_micro2<4>(g, gk, uh[pr], ul[pr]);
                    else if (st == 5) gelu_micro2<5>(g, gk, uh[pr], ul[pr]);
                    else if (st == 6) gelu_micro2<6>(g, gk, uh[pr], ul[pr]);
                    else gelu_micro2<7>(g, gk, uh[pr], ul[pr]);
                }
                if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc ^= uh[i] + ul[i]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) Xv[i] += __builtin_bit_cast(float, (acc & 0x7fffu) | 0x3f000000u) * 1e-3f / sinv;      // keep the inputs changing
    }
    const unsigned long long t1 = stamp();
    out[blockIdx.x * blockDim.x + tid] = acc;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves, const float* in, unsigned* out, unsigned long long* cyc) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        k<MODE><<<blocks, waves * 64, 160 * 1024, 0>>>(in, out, cyc, iters, 1.0f / (2048.f * 16384.f), 1024.f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) c.push_back((double)h[b * 8 + w] / iters);
    std::sort(c.begin(), c.end());
    printf("%-44s %d waves/CU: %8.0f ticks per tile of 16 x 64 elements (240 GELU + ~30 other instructions) = %.2f ticks per GELU instruction; %.1f ns\n",
           name, waves, c[c.size() / 2], c[c.size() / 2] / 240.0, ms * 1e6 / iters);
}

int main() {
    float* in; unsigned* out; unsigned long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<float> h(4096);
    unsigned s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9 & 0xffff) - 32768) / 8192.0f; }
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0>("two pairs alternating, fenced per micro-step", 4, in, out, cyc);
    run<0>("two pairs alternating, fenced per micro-step", 8, in, out, cyc);
    run<1>("eight pairs abreast, fenced per step", 4, in, out, cyc);
    run<1>("eight pairs abreast, fenced per step", 8, in, out, cyc);
    run<2>("eight pairs abreast, compiler's order", 4, in, out, cyc);
    run<2>("eight pairs abreast, compiler's order", 8, in, out, cyc);
    return 0;
}
