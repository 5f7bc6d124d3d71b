// Stand-alone timing lab for the fused MLP kernel (diagnostic; not part of libacx).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../audioset-convnext-inf_amd/csrc/mlp_fused.hip"
namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
}
int main() {
    const long long P0 = 64LL * 252 * 56, P1 = P0 / 4;
    struct S { int C; long long M; } shapes[] = {{96, P0}, {192, P1}};
    float *y, *x, *w, *b1, *b2;
    hipMalloc(&y, P0 * 96 * 4); hipMalloc(&x, P0 * 96 * 4); hipMalloc(&w, (size_t)4 * 192 * 192 * 2 * 4);
    hipMalloc(&b1, 4 * 192 * 4); hipMalloc(&b2, 192 * 4);
    std::vector<float> h((size_t)P0 * 96);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((int)((i * 2654435761u) >> 8 & 0xffff) - 32768) / 32768.f;
    hipMemcpy(y, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)4 * 192 * 192 * 2; ++i) h[i] *= 0.05f;
    hipMemcpy(w, h.data(), (size_t)4 * 192 * 192 * 2 * 4, hipMemcpyHostToDevice);
    hipMemcpy(b1, h.data(), 4 * 192 * 4, hipMemcpyHostToDevice); hipMemcpy(b2, h.data(), 192 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& s : shapes) {
        acx::BlockW bw; bw.wpack = w; bw.b1 = b1; bw.b2 = b2;
        if (acx::launch_mlp_fused(nullptr, bw, s.C, y, x, s.M, 0) != 0) return 1;
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) acx::launch_mlp_fused(nullptr, bw, s.C, y, x, s.M, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double tf = 4.0 * s.M * s.C * 4 * s.C / (ms * 1e-3) / 1e12;
        printf("fused C=%-4d M=%-8lld %8.1f us  %6.1f TF (%4.1f%% of 157.3)\n", s.C, s.M, ms * 1e3, tf, 100 * tf / 157.3);
    }
    return 0;
}
