// Does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 subnormal INPUTS?  (MI200 flushed them; the split-fp16
// GEMM must know.)  Also checks v_permlane32_swap semantics used by its epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float* out, unsigned* sw) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }   // 2^-20: subnormal in fp16
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
    unsigned x = 100 + threadIdx.x, y = 200 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    sw[2 * threadIdx.x] = r[0]; sw[2 * threadIdx.x + 1] = r[1];
}
int main() {
    float* d; unsigned* s;
    hipMalloc(&d, 64 * 4); hipMalloc(&s, 128 * 4);
    probe<<<1, 64>>>(d, s);
    float h[64]; unsigned hs[128];
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 512, hipMemcpyDeviceToHost);
    printf("mfma f16 subnormal input: got %g, expected %g if honoured (0 if flushed)\n", h[0], 16 * 9.5367431640625e-07);
    printf("permlane32_swap(x=100+l, y=200+l): lane0 -> (%u,%u)  lane32 -> (%u,%u)\n", hs[0], hs[1], hs[64], hs[65]);
    return 0;
}
