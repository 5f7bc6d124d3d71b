// Stand-alone timing lab for the N-split fused bf16 MLP (mlp_nsplit_bf16.hip) beside the ring kernel (mlp_fused_wide_bf16.hip); not
// part of libacx:
//   hipcc -O3 -std=c++17 -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 --offload-arch=gfx950 -w -Iaudioset-convnext-inf_amd/csrc -Itools/lab \
//         [-DACX_NS_STAMPS] [-DACX_NS_NOGELU=1] [-DACX_NS_NOMFMA=1] tools/lab/ns_lab.hip -o /tmp/ns_lab
//   /tmp/ns_lab [M]      default M = stage 2's pixel count at B = 64 (56 448); "ring" as a second argument times the ring kernel
// -DACX_NS_STAMPS: prints the median s_memtime deltas between the marks of wave 0 over the first 64 workgroups.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "mlp_nsplit_bf16.hip"
#include "mlp_fused_wide_bf16.hip"

namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
void prof_next_events(hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
thread_local int tls_inflight_ways = 1;
Tuning& tuning() { static Tuning t; return t; }
}
int main(int argc, char** argv) {
    const int C = 384;
    const long long M = argc > 1 ? atoll(argv[1]) : 64LL * 63 * 14;
    const bool ring = argc > 2 && !strcmp(argv[2], "ring");
    void *y, *x; float *b1, *b2; char* w;
    const size_t wbytes = (size_t)2 * (4 * C / 64) * 128 * C;
    hipMalloc(&y, M * C * 2); hipMalloc(&x, M * C * 2); hipMalloc(&b1, 4 * C * 4); hipMalloc(&b2, C * 4); hipMalloc(&w, wbytes);
    {
        std::vector<float> h((size_t)4 * C);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((int)((i * 2654435761u) >> 20 & 0xfff) - 2048) * 1e-3f;
        hipMemcpy(b1, h.data(), 4 * C * 4, hipMemcpyHostToDevice);
        hipMemcpy(b2, h.data(), C * 4, hipMemcpyHostToDevice);
        std::vector<uint16_t> hb((size_t)M * C);
        for (size_t i = 0; i < hb.size(); ++i) { const float v = (float)((int)((i * 2654435761u) >> 20 & 0xfff) - 2048) * 1e-3f; uint32_t u; std::memcpy(&u, &v, 4); hb[i] = (uint16_t)(u >> 16); }
        hipMemcpy(y, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(x, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        std::vector<uint16_t> hw(wbytes / 2);
        for (size_t i = 0; i < hw.size(); ++i) { const float v = ((int)(((i * 2654435761u) >> 9) & 0x7ff) - 1024) * 1e-4f; uint32_t u; std::memcpy(&u, &v, 4); hw[i] = (uint16_t)(u >> 16); }
        hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    }
    acx::BlockW bw;
    bw.wstream_b = reinterpret_cast<uint16_t*>(w); bw.b1 = b1; bw.b2 = b2;
#define CALL() (ring ? acx::launch_mlp_fused_wide_bf16(nullptr, bw, C, y, x, M, 0, nullptr, 0, true) : acx::launch_mlp_nsplit_bf16(nullptr, bw, C, y, x, M, 0))
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) if (CALL() != 0) return 1;
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) CALL();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    const double tf = 4.0 * M * C * 4 * C / (best * 1e-3) / 1e12;
    printf("%s C=%d M=%lld: %.1f us per block, %.1f TFLOP/s = %.3f of 2500\n", ring ? "ring  " : "nsplit", C, M, best * 1e3, tf, tf / 2500.0);
#ifdef ACX_NS_STAMPS
    if (!ring) {
        using namespace acx;
        static unsigned long long st[kNsStampBlocks * kNsStampSlots];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(acx_ns_stamps), sizeof st);
        const int nblk = (int)std::min<long long>(kNsStampBlocks, (M + 127) / 128);
        printf("mark: median ticks since the previous mark over %d workgroups (wave 0); s_memtime ticks at 100 MHz\n", nblk);
        long long total = 0;
        for (int m = 1; m < kNsStampSlots; ++m) {
            std::vector<long long> d;
            for (int b = 0; b < nblk; ++b) { const unsigned long long* s = st + b * kNsStampSlots; if (s[m] && s[m - 1]) d.push_back((long long)(s[m] - s[m - 1])); }
            if (d.empty()) break;
            std::sort(d.begin(), d.end());
            total += d[d.size() / 2];
            printf("  %3d %7lld%s", m, d[d.size() / 2], m % 10 == 0 ? "\n" : "");
        }
        printf("\n  sum of medians: %lld ticks\n", total);
    }
#endif
    return 0;
}
