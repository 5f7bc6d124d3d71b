// Pure-register MFMA load generators (no LDS, no global traffic in the loop) to run next to LDS-using victims.
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int KIND>
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
    h8 a[4], b[4]; b8 ab[4], bb[4];
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = hashu(threadIdx.x * 64 + s * 8 + i + blockIdx.x * 7919);
            const float fa = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.f), fb = ((int)(h >> 16) - 32768) * (1.0f / 32768.f);
            a[s][i] = (_Float16)fa; b[s][i] = (_Float16)fb; ab[s][i] = (__bf16)fa; bb[s][i] = (__bf16)fb;
        }
    f16v c[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (KIND == 0) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u + n) & 3], c[n], 0, 0, 0);
                else c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[u & 3], bb[(u + n) & 3], c[n], 0, 0, 0);
            }
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
    if (s == 123.456f) out[threadIdx.x] = s;
}
extern "C" int burn_launch(int kind, float* out, int blocks, int iters, void* stream) {
    if (kind == 0) burn<0><<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters);
    else burn<1><<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters);
    return (int)hipGetLastError();
}
