// VGPR canary: every lane keeps NV values live in registers, sleeps, and re-checks them many times.
#include <hip/hip_runtime.h>
constexpr int NV = 96;
__global__ __launch_bounds__(256) void vcanary(unsigned* report, int rounds) {
    unsigned v[NV];
    const unsigned tag = blockIdx.x * 2654435761u + threadIdx.x * 40503u;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = tag + i * 0x9E3779B9u; asm volatile("" : "+v"(v[i])); }
    for (int r = 0; r < rounds; ++r) {
        __builtin_amdgcn_s_sleep(32);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            asm volatile("" : "+v"(v[i]));
            if (v[i] != tag + i * 0x9E3779B9u) {
                const unsigned slot = atomicAdd(&report[0], 1u);
                if (slot < 64) { report[4 + 4 * slot] = blockIdx.x; report[5 + 4 * slot] = i; report[6 + 4 * slot] = v[i]; report[7 + 4 * slot] = threadIdx.x; }
                v[i] = tag + i * 0x9E3779B9u;
            }
        }
    }
}
extern "C" int vcanary_launch(unsigned* report, int blocks, int rounds, void* stream) {
    vcanary<<<blocks, 256, 0, (hipStream_t)stream>>>(report, rounds);
    return (int)hipGetLastError();
}
