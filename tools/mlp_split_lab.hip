// Stand-alone timing lab for the fused split-fp16 MLP kernel (diagnostic; not part of libacx).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../audioset-convnext-inf_amd/csrc/mlp_fused_split.hip"
namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
}
int main() {
    const long long P0 = 64LL * 252 * 56, P1 = P0 / 4;
    struct S { int C; long long M; } shapes[] = {{96, P0}, {192, P1}
#ifdef ACX_FSLAB_MALL     // whole rounds of 512 workgroups: 13, 3 and 1 -- does a cache-resident activation set run faster per round?
        , {96, 512LL * 128 * 13}, {96, 512LL * 128 * 3}, {96, 512LL * 128}, {192, 512LL * 128 * 3}, {192, 512LL * 128}
#endif
    };
    float *y, *x, *b1, *b2; uint16_t* w;
    hipMalloc(&y, P0 * 96 * 4); hipMalloc(&x, P0 * 96 * 4); hipMalloc(&w, (size_t)4 * 192 * 192 * 2 * 4);
    hipMalloc(&b1, 4 * 192 * 4); hipMalloc(&b2, 192 * 4);
    std::vector<float> h((size_t)P0 * 96);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((int)((i * 2654435761u) >> 8 & 0xffff) - 32768) / 32768.f;
    hipMemcpy(y, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b1, h.data(), 4 * 192 * 4, hipMemcpyHostToDevice); hipMemcpy(b2, h.data(), 192 * 4, hipMemcpyHostToDevice);
    {
        std::vector<uint16_t> hw((size_t)4 * 192 * 192 * 2 * 2);
        for (size_t i = 0; i < hw.size(); ++i) {
            const unsigned r = (unsigned)((i * 2654435761u) >> 9);
            const bool lo = (i >> 3) & 1;
            const _Float16 v = (_Float16)(((int)(r & 0x7ff) - 1024) * (lo ? 1e-3f : 8.f));
            std::memcpy(&hw[i], &v, 2);
        }
        hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& s : shapes) {
        acx::BlockW bw; bw.wpack_s = w; bw.b1 = b1; bw.b2 = b2; bw.w1s_scale = 16384.f; bw.w2s_scale = 16384.f;
        if (acx::launch_mlp_fused_split(nullptr, bw, s.C, y, x, s.M, 0) != 0) return 1;
        hipDeviceSynchronize();
        float ms = 1e30f;
        for (int batch = 0; batch < 5; ++batch) {       // min of 5 batches of 10 launches (boxes drift)
            hipEventRecord(e0, 0);
            for (int r = 0; r < 10; ++r) acx::launch_mlp_fused_split(nullptr, bw, s.C, y, x, s.M, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); t /= 10;
            ms = t < ms ? t : ms;
        }
        double tf = 4.0 * s.M * s.C * 4 * s.C / (ms * 1e-3) / 1e12;
        printf("fused-split C=%-4d M=%-8lld %8.1f us  %6.1f TF fp32-equivalent (%4.1f%% of 833)  %6.2f us per round of 512 workgroups\n", s.C, s.M, ms * 1e3, tf, 100 * tf / 833.3, ms * 1e3 / (s.M / (512.0 * 128)));
#ifdef ACX_FSLAB_STAMP
        unsigned long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(acx::acx_fs_stamps), sizeof(st));
        const double nw = (double)st[6], nc = nw * (4 * s.C / 32);
        printf("   per chunk (s_memtime ticks): phase1+GELU %.0f | phase2 %.0f | barrier %.0f   || per wave: prologue %.0f, loop %.0f, epilogue %.0f\n",
               st[0] / nc, st[1] / nc, st[2] / nc, st[3] / nw, st[4] / nw, st[5] / nw);
        printf("   wave lifetime: %.0f s_memtime ticks = %.2f us of s_memrealtime (100 MHz) -> %.3f ticks per ns\n",
               (st[3] + st[4] + st[5]) / nw, st[7] / nw / 1000.0 / 100.0, (double)(st[3] + st[4] + st[5]) / (st[7] / 1000.0 * 10.0));
        unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(acx::acx_fs_stamps), z, sizeof(z));
#endif
    }
    return 0;
}
