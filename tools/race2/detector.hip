// Sensitive pure-register detector (the radix-8 butterfly chain that went wrong in session 2), built twice:
// default flags (hipcc turns the complex arithmetic into packed-FP32 v_pk_* instructions) and -fno-slp-vectorize
// (32-bit v_fma / v_mul / v_add only).  No LDS, no memory traffic inside the loop.
#include <hip/hip_runtime.h>
#include "../../audioset-convnext-inf_amd/csrc/fft_core.h"
using namespace acx;
__global__ __launch_bounds__(256) void detector_kernel(const float* __restrict__ seed, float* __restrict__ out, int rounds) {
    extern __shared__ float unused_lds[];      // only sized by the launch: keeps the workgroup off CUs whose LDS is taken
    if (rounds < 0) unused_lds[threadIdx.x] = 0.f;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    cf v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = cf_make(seed[(gid * 16 + 2 * r) & 0xffff], seed[(gid * 16 + 2 * r + 1) & 0xffff]);
    const cf w = cf_make(0.98078528f, -0.19509032f);
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cf_mul(v[r], w);
        fft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r].x *= 0.35355339f; v[r].y *= 0.35355339f; }
    }
    float c = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) c += v[r].x + 3.f * v[r].y;
    out[gid] = c;
}
// lds_bytes > 0: the workgroup claims that much (unused) LDS -- with 160 KB it can only run on a CU that holds no other
// LDS-using workgroup, i.e. never next to a GEMM workgroup
extern "C" int detector_launch_lds(const float* seed, float* out, int blocks, int rounds, int lds_bytes, void* stream) {
    if (lds_bytes > 0 && hipFuncSetAttribute(reinterpret_cast<const void*>(&detector_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -2;
    detector_kernel<<<dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream>>>(seed, out, rounds);
    return (int)hipGetLastError();
}
extern "C" int detector_launch(const float* seed, float* out, int blocks, int rounds, void* stream) {
    detector_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(seed, out, rounds);
    return (int)hipGetLastError();
}
