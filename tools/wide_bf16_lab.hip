// Stand-alone timing lab for the fused bf16 MLP kernel (not part of libacx):
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -w -DWIDE_SRC='"path/to/mlp_fused_wide_bf16.hip"' -DWIDE_C=96 tools/wide_bf16_lab.hip -o /tmp/wide_bf16_lab
// tools/run_wide_bf16_lab.sh builds sed-patched variants of the product source (no DMA, no GELU, ...) and runs them all.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include WIDE_SRC

namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
}

int main(int argc, char** argv) {
#ifndef WIDE_C
#define WIDE_C 384
#endif
    const int C = WIDE_C;
    const long long M = argc > 1 ? atoll(argv[1]) : 64LL * 252 * 56 * 96 / C;
    float *y, *x, *b1, *b2; char* w;
    const size_t wbytes = (size_t)2 * 4 * C * C * 2;
    hipMalloc(&y, M * C * 4); hipMalloc(&x, M * C * 4); hipMalloc(&b1, 4 * C * 4); hipMalloc(&b2, C * 4); hipMalloc(&w, wbytes);
    {
        std::vector<float> h((size_t)M * C);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((int)((i * 2654435761u) >> 20 & 0xfff) - 2048) * 1e-3f;
        hipMemcpy(y, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(b1, h.data(), 4 * C * 4, hipMemcpyHostToDevice);
        hipMemcpy(b2, h.data(), C * 4, hipMemcpyHostToDevice);
        std::vector<uint16_t> hw(wbytes / 2);
        for (size_t i = 0; i < hw.size(); ++i) {
            const unsigned r = (unsigned)((i * 2654435761u) >> 9);
            const float v = ((int)(r & 0x7ff) - 1024) * 3e-5f;
            unsigned u; std::memcpy(&u, &v, 4);
            hw[i] = (uint16_t)(u >> 16);
        }
        hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    }
    acx::BlockW bw;
    bw.wstream_b = reinterpret_cast<uint16_t*>(w); bw.b1 = b1; bw.b2 = b2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) if (acx::launch_mlp_fused_wide_bf16(nullptr, bw, C, y, x, M, 0, nullptr, 0) != 0) return 1;
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) acx::launch_mlp_fused_wide_bf16(nullptr, bw, C, y, x, M, 0, nullptr, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    const double tf = 4.0 * M * C * 4 * C / (best * 1e-3) / 1e12;
    printf("C=%d M=%lld: %.1f us per block, %.1f TF = %.3f of 2500\n", C, M, best * 1e3, tf, tf / 2500.0);
    return 0;
}
