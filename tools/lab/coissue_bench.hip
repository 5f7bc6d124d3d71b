// Micro-benchmark: do fp32 vector instructions of one wave run beside the MFMAs of another wave of the same SIMD -- and
// beside the MFMAs of the SAME wave?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/coissue_bench.hip -o /tmp/coissue_bench && /tmp/coissue_bench
// One workgroup per CU (160 KB LDS requested), 4 or 8 waves; per loop iteration a wave issues NM MFMAs
// (v_mfma_f32_32x32x16_f16, two accumulators alternating) and / or NV v_fma_f32 on 8 independent registers.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ROLE: 0 every wave MFMA only, 1 every wave VALU only, 2 every wave both (interleaved 1 MFMA : NV/NM FMAs),
//       3 waves 0-3 MFMA only, waves 4-7 VALU only (needs 8 waves)
template <int ROLE, int NM, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = ROLE == 0 || ROLE == 2 || (ROLE == 3 && wave < 4);
    const bool do_v = ROLE == 1 || ROLE == 2 || (ROLE == 3 && wave >= 4);
    f32x16 acc0 = {}, acc1 = {};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const float c1 = 1.0001f, c2 = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (ROLE == 2) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], c1, c2);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                }
            }
            if (do_v) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], c1, c2);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ROLE, int NM, int NV>
float run(float* out, int waves, const char* name) {
    const int iters = 20000;
    hipFuncSetAttribute((const void*)k<ROLE, NM, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ROLE, NM, NV><<<256, waves * 64, 160 * 1024>>>(out, 100);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<ROLE, NM, NV><<<256, waves * 64, 160 * 1024>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-58s %d waves: %8.1f ns per iteration\n", name, waves, best * 1e6 / iters);
    return best;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    printf("per iteration and wave: 16 MFMAs (512 cycles of the matrix pipe) and / or 96 v_fma_f32 (384 issue cycles)\n");
    run<0, 16, 96>(out, 4, "MFMA only");
    run<1, 16, 96>(out, 4, "VALU only");
    run<2, 16, 96>(out, 4, "same wave: 1 MFMA : 6 FMAs interleaved");
    run<0, 16, 96>(out, 8, "MFMA only");
    run<1, 16, 96>(out, 8, "VALU only");
    run<2, 16, 96>(out, 8, "same wave: 1 MFMA : 6 FMAs interleaved");
    run<3, 16, 96>(out, 8, "waves 0-3 MFMA only, waves 4-7 VALU only");
    run<2, 16, 192>(out, 4, "same wave: 1 MFMA : 12 FMAs interleaved");
    run<1, 16, 192>(out, 4, "VALU only, 192 FMAs");
    run<2, 16, 32>(out, 4, "same wave: 1 MFMA : 2 FMAs interleaved");
    return 0;
}
