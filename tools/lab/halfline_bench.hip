// Lab: what does a 64-byte half of a 128-byte line cost against the whole line?  A tensor of P pixels x 192 B (NHWC bf16, C = 96:
// three 64-byte slices per pixel, as the bf16 depthwise kernels see stage 0) or x 256 B is written, read, or copied by waves that
// each own ONE slice of SEG bytes per pixel: SEG = 64 (neighbouring waves of a workgroup own the two halves of a line) or SEG = 128.
// Same bytes, same instruction width (16 B per lane), same number of waves.
//   hipcc -O3 --offload-arch=gfx950 tools/lab/halfline_bench.hip -o build/lab/halfline_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0 write, 1 read, 2 copy.  A wave-instruction covers 64 * 16 / SEG pixels of the wave's slice.
template <int SEG, int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, char* __restrict__ dst, int pitch, long long pixels_per_wave, float* sinkp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kLanesPerSeg = SEG / 16, kPxPerInstr = 64 / kLanesPerSeg;
    const int nsl = pitch / SEG;
    const long long item = (long long)blockIdx.x * 4 + wave;
    const int slice = (int)(item % nsl);
    const long long p0 = (item / nsl) * pixels_per_wave;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long lane_off = (long long)(lane / kLanesPerSeg) * pitch + slice * SEG + (lane % kLanesPerSeg) * 16;
    for (long long p = 0; p < pixels_per_wave; p += kPxPerInstr * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long off = (p0 + p + u * kPxPerInstr) * pitch + lane_off;
            if (MODE == 0) *reinterpret_cast<f4*>(dst + off) = f4{(float)p, (float)u, 1.f, 2.f};
            else if (MODE == 1) { const f4 v = *reinterpret_cast<const f4*>(src + off); acc += v; }
            else *reinterpret_cast<f4*>(dst + off) = *reinterpret_cast<const f4*>(src + off);
        }
    }
    if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sinkp[0] = acc[0];
}

int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* sinkp; hipMalloc(&sinkp, 64);
    for (int pitch : {192, 256, 384}) {
        const long long P = 64LL * 252 * 56 * 192 / pitch;        // 173 MB
        const size_t bytes = (size_t)P * pitch;
        char *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
        for (int seg : {64, 128}) {
            if (pitch % seg) continue;
            const int nsl = pitch / seg;
            for (int mode = 0; mode < 3; ++mode)
                for (int waves : {2048, 4096}) {
                    long long ppw = (P * nsl / waves + 63) / 64 * 64;       // pixels per wave (a multiple of the unroll)
                    const long long groups = (P + ppw - 1) / ppw;
                    const int blocks = (int)((groups * nsl + 3) / 4);
                    if ((groups * ppw) > P) { /* the last group would run past the end */ ppw = P / groups / 64 * 64; }
                    float best = 1e30f;
                    for (int rep = 0; rep < 5; ++rep) {
                        hipEventRecord(e0, 0);
#define LAUNCH(SEG_, MODE_) k<SEG_, MODE_><<<blocks, 256>>>(a, b, pitch, ppw, sinkp)
                        if (seg == 64) { if (mode == 0) LAUNCH(64, 0); else if (mode == 1) LAUNCH(64, 1); else LAUNCH(64, 2); }
                        else { if (mode == 0) LAUNCH(128, 0); else if (mode == 1) LAUNCH(128, 1); else LAUNCH(128, 2); }
                        hipEventRecord(e1, 0); hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1);
                        best = std::min(best, ms);
                    }
                    const double moved = (double)groups * ppw * pitch * (mode == 2 ? 2 : 1);
                    printf("pitch %3d B  slice %3d B  %-5s  %4d waves  %7.1f us  %5.2f TB/s\n", pitch, seg, mode == 0 ? "write" : mode == 1 ? "read" : "copy",
                           blocks * 4, best * 1e3, moved / (best * 1e-3) / 1e12);
                }
        }
        hipFree(a); hipFree(b);
    }
    return 0;
}
