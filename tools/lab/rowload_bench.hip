// Micro-benchmark: how fast can 8 free-running waves per CU stream "row per lane" tiles (the operand layout of a 32x32x16 MFMA:
// lane (px, half) owns 8 consecutive channels per k-step) compared with fully coalesced accesses?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/rowload_bench.hip -o /tmp/rowload_bench && /tmp/rowload_bench
// Each kernel reads y and x (M x C fp32), reduces y per row (so the loads stay), writes x + f(y) back: 3 tensor passes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int C, int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ y, float* __restrict__ x, long long M) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const long long ntiles = M / 32;
    for (long long tile = (long long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long long)gridDim.x * 8) {
        float4 a[C / 8], b[C / 8];
        if (MODE == 0) {            // MFMA k-step layout: per k-step 16 s + 8 hh .. + 7  (two float4 per step, 64 B stride between steps)
            const float* yp = y + (tile * 32 + l31) * C + 8 * hh;
            const float* xp = x + (tile * 32 + l31) * C + 8 * hh;
#pragma unroll
            for (int s = 0; s < C / 16; ++s) { a[2 * s] = *(const float4*)(yp + 16 * s); a[2 * s + 1] = *(const float4*)(yp + 16 * s + 4); }
#pragma unroll
            for (int s = 0; s < C / 16; ++s) { b[2 * s] = *(const float4*)(xp + 16 * s); b[2 * s + 1] = *(const float4*)(xp + 16 * s + 4); }
        } else if (MODE == 1) {     // half a row per lane, contiguous
            const float* yp = y + (tile * 32 + l31) * C + (C / 2) * hh;
            const float* xp = x + (tile * 32 + l31) * C + (C / 2) * hh;
#pragma unroll
            for (int i = 0; i < C / 8; ++i) a[i] = *(const float4*)(yp + 4 * i);
#pragma unroll
            for (int i = 0; i < C / 8; ++i) b[i] = *(const float4*)(xp + 4 * i);
        } else {                    // fully coalesced: instruction i covers 1 KB
            const float* yp = y + tile * 32 * C + 4 * lane;
            const float* xp = x + tile * 32 * C + 4 * lane;
#pragma unroll
            for (int i = 0; i < C / 8; ++i) a[i] = *(const float4*)(yp + 256 * i);
#pragma unroll
            for (int i = 0; i < C / 8; ++i) b[i] = *(const float4*)(xp + 256 * i);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < C / 8; ++i) sum += (a[i].x + a[i].y) + (a[i].z + a[i].w);
        sum += __shfl_xor(sum, 32);
        sum *= 1e-9f;
#pragma unroll
        for (int i = 0; i < C / 8; ++i) { b[i].x += sum; b[i].y += sum; b[i].z += sum; b[i].w += sum; }
        if (MODE == 0) {
            float* xp = x + (tile * 32 + l31) * C + 8 * hh;
#pragma unroll
            for (int s = 0; s < C / 16; ++s) { *(float4*)(xp + 16 * s) = b[2 * s]; *(float4*)(xp + 16 * s + 4) = b[2 * s + 1]; }
        } else if (MODE == 1) {
            float* xp = x + (tile * 32 + l31) * C + (C / 2) * hh;
#pragma unroll
            for (int i = 0; i < C / 8; ++i) *(float4*)(xp + 4 * i) = b[i];
        } else {
            float* xp = x + tile * 32 * C + 4 * lane;
#pragma unroll
            for (int i = 0; i < C / 8; ++i) *(float4*)(xp + 256 * i) = b[i];
        }
    }
}

template <int C, int MODE>
void run(const float* y, float* x, long long M, int wgs, size_t lds, const char* name) {
    hipFuncSetAttribute((const void*)k<C, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) k<C, MODE><<<wgs, 512, lds>>>(y, x, M);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) k<C, MODE><<<wgs, 512, lds>>>(y, x, M);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    printf("C=%3d %-28s wgs %5d lds %6zu: %.1f us  %.2f TB/s\n", C, name, wgs, lds, best * 1e3, 3.0 * M * C * 4 / (best * 1e-3) / 1e12);
}

template <int C>
void all(long long M) {
    float *y, *x;
    hipMalloc(&y, M * C * 4); hipMalloc(&x, M * C * 4);
    hipMemset(y, 0, M * C * 4); hipMemset(x, 0, M * C * 4);
    for (int wgs : {256, 512, 2048}) {
        const size_t lds = wgs == 256 ? 160 * 1024 : (wgs == 512 ? 80 * 1024 : 1024);
        run<C, 0>(y, x, M, wgs, lds, "k-step layout (product)");
        run<C, 1>(y, x, M, wgs, lds, "half row per lane");
        run<C, 2>(y, x, M, wgs, lds, "coalesced");
    }
    hipFree(y); hipFree(x);
}

int main() {
    all<96>(64LL * 252 * 56);
    all<192>(64LL * 126 * 28);
    return 0;
}
