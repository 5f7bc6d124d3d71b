// Stand-alone timing lab for the depthwise 7x7 kernel (diagnostic; not part of libacx).
#include <cstdio>
#include <vector>
#ifdef ACX_LAB_DW_V1
#include "experimental/dwconv_v1.hip"
#else
#include "../audioset-convnext-inf_amd/csrc/dwconv.hip"
#endif
namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
}
int main() {
    struct S { int C, H, W; } shapes[] = {{96, 252, 56}, {192, 126, 28}, {384, 63, 14}, {768, 31, 7}};
    const int B = 64;
    const size_t n = (size_t)B * 252 * 56 * 96;
    float *x, *y, *w, *bias;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, 49 * 768 * 4); hipMalloc(&bias, 768 * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((int)((i * 2654435761u) >> 8 & 0xffff) - 32768) / 32768.f;
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), 49 * 768 * 4, hipMemcpyHostToDevice); hipMemcpy(bias, h.data(), 768 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& s : shapes) {
        acx::BlockW bw; bw.dw = w; bw.dwb = bias;
        if (acx::launch_dwconv(nullptr, bw, s.C, x, y, nullptr, B, s.H, s.W, 0) != 0) return 1;
        hipDeviceSynchronize();
        float best = 1e30f, sum = 0.f;
        for (int batch = 0; batch < 6; ++batch) {          // min and mean of 6 batches of 20 launches: boxes drift by +-10 %
            hipEventRecord(e0, 0);
            for (int r = 0; r < 20; ++r) acx::launch_dwconv(nullptr, bw, s.C, x, y, nullptr, B, s.H, s.W, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); t /= 20;
            best = t < best ? t : best; sum += t;
        }
        float ms = best;
        double bytes = 2.0 * B * s.H * s.W * s.C * 4;
        printf("dwconv C=%-4d %3dx%-3d %8.1f us (mean %6.1f)  %6.2f TB/s algorithmic (%4.1f%% of 8 TB/s)\n", s.C, s.H, s.W, ms * 1e3, sum / 6 * 1e3, bytes / (ms * 1e-3) / 1e12, 100 * bytes / (ms * 1e-3) / 8e12);
#ifdef ACX_LAB_DW_STAMP
        unsigned long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(acx::acx_dw_stamps), sizeof(st));
        double n = (double)st[5];
        printf("   per wave-tile cycles: load-issue+FMA %.0f | store-issue %.0f | barrier1 %.0f | vmcnt+ds_write %.0f | barrier2 %.0f   (tiles %.0f)\n", st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, n);
        unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(acx::acx_dw_stamps), z, sizeof(z));
#endif
    }
    return 0;
}
