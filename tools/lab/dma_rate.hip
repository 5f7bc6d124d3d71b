// Lab (not part of libacx): how fast can ONE CU pull an L2-resident weight stream into its LDS?
//   mode 0  global_load_lds_dwordx4 (LDS-DMA), 1-KB pieces, `depth` pieces in flight per wave
//   mode 1  global_load_dwordx4 into registers only (the L2 -> L1 -> VGPR path, no LDS write)
//   mode 2  global_load_dwordx4 into registers + ds_write_b128 (register-staged fill)
// Every workgroup (one per CU, `waves` waves) streams the SAME buffer of `kb` KB `reps` times, as the fused MLP kernels stream their
// weights: the bytes come from the L2 / Infinity Cache, not from HBM.
//   hipcc -O3 --offload-arch=gfx950 tools/lab/dma_rate.hip -o build/labs/dma_rate && build/labs/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ src, int pieces_per_wave, int reps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    float4 accv = {0.f, 0.f, 0.f, 0.f};
    char* lds_wave = smem + wave * (DEPTH * 1024);
    for (int r = 0; r < reps; ++r) {
        const char* base = src + (size_t)wave * pieces_per_wave * 1024 + lane * 16;
        for (int p0 = 0; p0 < pieces_per_wave; p0 += DEPTH) {
            if (MODE == 0) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(p0 + d) * 1024),
                                                     (__attribute__((address_space(3))) void*)(lds_wave + d * 1024), 16, 0, 0);
                // half of the group may stay in flight while the next group is issued
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH / 2) : "memory");
            } else {
                float4 v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const float4*>(base + (size_t)(p0 + d) * 1024);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    if (MODE == 2) *reinterpret_cast<float4*>(lds_wave + d * 1024 + lane * 16) = v[d];
                    else { accv.x += v[d].x; accv.y += v[d].w; }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 1) accv.x += *reinterpret_cast<float*>(smem + threadIdx.x * 4);
    if (accv.x == 12345.678f) sink[0] = accv.x + accv.y;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, int kb, int waves, int reps, float* sink, int cus) {
    const int pieces_per_wave = kb / waves / DEPTH * DEPTH;
    const size_t lds = (size_t)waves * DEPTH * 1024 + 2048;
    hipFuncSetAttribute((const void*)&stream_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), dim3(cus), dim3(waves * 64), 160 * 1024 /* one workgroup per CU */, 0, src, pieces_per_wave, reps, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)lds;
    const double bytes_cu = (double)pieces_per_wave * waves * 1024.0 * reps;
    printf("%-34s depth %2d waves %d: %7.1f GB/s per CU  = %5.1f B/clk at 2.1 GHz, chip %6.2f TB/s  (%.3f ms)\n", name, DEPTH, waves,
           bytes_cu / (best * 1e-3) / 1e9, bytes_cu / (best * 1e-3) / 2.1e9, bytes_cu * cus / (best * 1e-3) / 1e12, best);
}

int main(int argc, char** argv) {
    const int kb = argc > 1 ? atoi(argv[1]) : 2304;       // the C = 384 bf16 weight stream of one block
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    char* src; float* sink;
    hipMalloc(&src, (size_t)kb * 1024 + 65536); hipMemset(src, 1, (size_t)kb * 1024 + 65536); hipMalloc(&sink, 64);
    printf("stream of %d KB read %d times by each of %d CUs\n", kb, reps, cus);
    for (int waves : {4, 8}) {
        run<0, 4>("LDS-DMA", src, kb, waves, reps, sink, cus);
        run<0, 8>("LDS-DMA", src, kb, waves, reps, sink, cus);
        run<0, 16>("LDS-DMA", src, kb, waves, reps, sink, cus);
        run<1, 8>("global_load -> registers", src, kb, waves, reps, sink, cus);
        run<1, 16>("global_load -> registers", src, kb, waves, reps, sink, cus);
        run<2, 8>("global_load + ds_write_b128", src, kb, waves, reps, sink, cus);
        run<2, 16>("global_load + ds_write_b128", src, kb, waves, reps, sink, cus);
    }
    return 0;
}
