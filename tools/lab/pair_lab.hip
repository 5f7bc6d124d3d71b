// Stand-alone timing lab for the paired fused bf16 MLP (mlp_pair_bf16.hip; not part of libacx):
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -w -DPAIR_C=384 -Iaudioset-convnext-inf_amd/csrc -DPAIR_SRC='"mlp_pair_bf16.hip"' -Itools/lab tools/lab/pair_lab.hip -o /tmp/pair_lab
//   /tmp/pair_lab [M]      default M = the stage's pixel count at B = 64
// -DACX_PAIR_STAMPS: prints, for the first tile of the first 64 workgroups, the median cycles of every interval of producer wave 0 and
// consumer wave 4: [work before the mark | gelu or nothing | MFMA phase | wait + barrier].
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include PAIR_SRC

namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
void prof_next_events(hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
thread_local int tls_inflight_ways = 1;
Tuning& tuning() { static Tuning t; return t; }
}
#ifndef PAIR_C
#define PAIR_C 384
#endif
int main(int argc, char** argv) {
    const int C = PAIR_C;
    const long long M = argc > 1 ? atoll(argv[1]) : 64LL * 252 * 56 * 96 / C;
    void *y, *x; float *b1, *b2; char* w;
    const size_t wbytes = (size_t)2 * (4 * C / 32) * 64 * C;
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
#define CALL() acx::launch_mlp_pair_bf16(nullptr, bw, C, y, x, M, 0, nullptr, 0, true)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) if (CALL() != 0) return 1;
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) CALL();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    const double tf = 4.0 * M * C * 4 * C / (best * 1e-3) / 1e12;
    printf("C=%d M=%lld: %.1f us per block, %.1f TFLOP/s = %.3f of 2500\n", C, M, best * 1e3, tf, tf / 2500.0);
#ifdef ACX_PAIR_STAMPS
    {
        using namespace acx;
        static unsigned long long st[kPairStampBlocks * 2 * kPairStampSlots];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(acx_pair_stamps), sizeof st);
        const int n = 4 * C / 32, marks = 3 * (n + 1);
        for (int role = 0; role < 2; ++role) {
            printf("%s: interval: [top->mark1, mark1->mark2, wait+barrier] median cycles over %d workgroups\n", role ? "consumer (wave 4)" : "producer (wave 0)", kPairStampBlocks);
            double tot[3] = {0, 0, 0};
            for (int k = 0; k <= n; ++k) {
                std::vector<long long> a, b, c;
                for (int blk = 0; blk < kPairStampBlocks; ++blk) {
                    const unsigned long long* s = st + (blk * 2 + role) * kPairStampSlots;
                    if (3 * k + 3 >= kPairStampSlots || s[3 * k + 2] == 0) continue;
                    a.push_back((long long)(s[3 * k + 1] - s[3 * k])); b.push_back((long long)(s[3 * k + 2] - s[3 * k + 1]));
                    if (3 * k + 3 < marks) c.push_back((long long)(s[3 * k + 3] - s[3 * k + 2]));
                }
                auto med = [](std::vector<long long>& v) { if (v.empty()) return 0LL; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
                const long long ma = med(a), mb = med(b), mc = med(c);
                if (k < 6 || k > n - 4 || k % 8 == 0) printf("  k=%2d  %6lld %6lld %6lld\n", k, ma, mb, mc);
                if (k >= 2 && k < n) { tot[0] += ma; tot[1] += mb; tot[2] += mc; }
            }
            printf("  mean over k = 2 .. n-1: %.0f %.0f %.0f  (sum %.0f per interval; MFMA floor of an interval: %d cycles per SIMD)\n", tot[0] / (n - 2), tot[1] / (n - 2), tot[2] / (n - 2),
                   (tot[0] + tot[1] + tot[2]) / (n - 2), 2 * 24 * 32);
        }
    }
#endif
    return 0;
}
