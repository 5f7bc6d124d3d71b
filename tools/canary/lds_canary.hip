// LDS canary, write/read-back flavour: each round every thread writes fresh values (tag + round + index) to the
// workgroup's LDS, barrier, reads them back through a permuted index, barrier.  Mismatches are reported with the
// value found, so one can tell a lost write (previous round's value) from a foreign write (somebody else's data).
#include <hip/hip_runtime.h>
#include <cstdint>
constexpr int kWords = 12 * 1024;        // 48 KB per workgroup
__global__ __launch_bounds__(256) void canary(unsigned* report, int rounds) {
    __shared__ unsigned buf[kWords];
    const unsigned tag = (blockIdx.x * 2654435761u) & 0xffff0000u;
    for (int s = 0; s < rounds; ++s) {
        for (int i = threadIdx.x; i < kWords; i += 256) buf[i] = tag + (unsigned)(s << 16 >> 16 << 0) * 0u + ((s & 0xf) << 12) + (i & 0xfff) + ((unsigned)(i >> 12) << 28 >> 28 << 0) * 0u;
        __syncthreads();
        for (int i = threadIdx.x; i < kWords; i += 256) {
            const int j = (i * 97 + 31) % kWords;
            const unsigned v = buf[j];
            const unsigned expect = tag + ((s & 0xf) << 12) + (j & 0xfff);
            if (v != expect) {
                const unsigned slot = atomicAdd(&report[0], 1u);
                if (slot < 64) { report[4 + 4 * slot] = blockIdx.x; report[5 + 4 * slot] = j; report[6 + 4 * slot] = v; report[7 + 4 * slot] = expect; }
            }
        }
        __syncthreads();
    }
}
extern "C" int canary_launch(unsigned* report, int blocks, int rounds, void* stream) {
    canary<<<blocks, 256, 0, (hipStream_t)stream>>>(report, rounds);
    return (int)hipGetLastError();
}
