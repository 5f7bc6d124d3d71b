// Fills the LDS of every CU with a bit pattern (launch before a kernel under test to expose reads of
// uninitialised LDS: 0x7fc00000 = NaN, 0x7f7f7f7f = 3.4e38, 0x77007700 = what split-fp16 operands look like).
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void poison(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned buf[];
    for (int i = threadIdx.x; i < 20480; i += 256) buf[i] = pattern;      // 80 KB
    __syncthreads();
    if (buf[(threadIdx.x * 7) % 20480] == 0x12345678u) sink[0] = 1;
}
extern "C" int poison_launch(unsigned pattern, unsigned* sink, void* stream) {
    static bool set = false;
    if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(&poison), hipFuncAttributeMaxDynamicSharedMemorySize, 81920); set = true; }
    poison<<<2048, 256, 81920, (hipStream_t)stream>>>(pattern, sink);
    return (int)hipGetLastError();
}
