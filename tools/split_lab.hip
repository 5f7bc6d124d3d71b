// Stand-alone timing lab for the split-fp16 GEMM kernel (diagnostic builds; not part of libacx).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DACX_SLAB_xxx] tools/split_lab.hip -o /tmp/split_lab
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../audioset-convnext-inf_amd/csrc/gemm_split.hip"
#ifdef ACX_LAB_WS
#include "experimental/gemm_split_ws.hip"
#endif

namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
}

int main(int argc, char** argv) {
    struct Shape { const char* name; long long M; int N, K; int epi; };
    const long long P0 = 64LL * 252 * 56, P1 = P0 / 4, P2 = P1 / 4, P3 = 64LL * 31 * 7;
    std::vector<Shape> shapes = {
        {"s0.pw1", P0, 384, 96, acx::EPI_GELU},   {"s0.pw2", P0, 96, 384, acx::EPI_RESID},
        {"s1.pw1", P1, 768, 192, acx::EPI_GELU},  {"s1.pw2", P1, 192, 768, acx::EPI_RESID},
        {"s2.pw1", P2, 1536, 384, acx::EPI_GELU}, {"s2.pw2", P2, 384, 1536, acx::EPI_RESID},
        {"s3.pw1", P3, 3072, 768, acx::EPI_GELU}, {"s3.pw2", P3, 768, 3072, acx::EPI_RESID},
    };
    size_t maxA = 0, maxO = 0;
    for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxO = std::max(maxO, (size_t)s.M * s.N); }
    char *A, *W, *O; float* bias;
    hipMalloc(&A, maxA * 4); hipMalloc(&O, maxO * 4); hipMalloc(&W, (size_t)3072 * 768 * 4); hipMalloc(&bias, 3072 * 4);
    {
        std::vector<uint16_t> h(maxA * 2);      // plausible S16 content: hi = small fp16 values, lo = tiny ones
        for (size_t i = 0; i < maxA * 2; ++i) {
            const unsigned r = (unsigned)((i * 2654435761u) >> 9);
            const bool lo = (i >> 3) & 1;
            const _Float16 v = (_Float16)(((int)(r & 0x7ff) - 1024) * (lo ? 1e-4f : 0.01f));
            std::memcpy(&h[i], &v, 2);
        }
        hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)3072 * 768 * 4, hipMemcpyHostToDevice);
        std::vector<float> f(maxO, 0.25f);
        hipMemcpy(O, f.data(), maxO * 4, hipMemcpyHostToDevice);
        hipMemcpy(bias, f.data(), 3072 * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 5;
    double total = 0;
    for (auto& s : shapes) {
        acx::GemmSplitArgs g{};
        g.A = A; g.Wt = W; g.bias = bias; g.out = O; g.resid = s.epi == acx::EPI_RESID ? (const float*)O : nullptr;
        g.M = s.M; g.N = s.N; g.K = s.K; g.sinv = 1e-3f; g.epi = s.epi; g.cls = 0;
#ifdef ACX_LAB_WS
        const bool ws = acx::gemm_split_ws_supported(g);
        auto launch = [&]() { return ws ? acx::launch_gemm_split_ws(nullptr, g, 0) : acx::launch_gemm_split(nullptr, g, 0); };
#else
        const bool ws = false;
        auto launch = [&]() { return acx::launch_gemm_split(nullptr, g, 0); };
#endif
        if (launch() != 0) return 1;
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        double tf = 2.0 * s.M * s.N * s.K / (ms * 1e-3) / 1e12;
        const int blocks = s.name[1] == '0' || s.name[1] == '1' || s.name[1] == '3' ? 3 : 9;
        total += ms * blocks;
        printf("%-8s%s M=%-8lld N=%-5d K=%-5d  %8.1f us  %6.1f TF fp32-equivalent (%4.1f%% of 833 = fp16 peak / 3)\n", s.name, ws ? " [ws]" : "", s.M, s.N, s.K,
               ms * 1e3, tf, 100 * tf / 833.3);
    }
    printf("all 18 blocks: %.2f ms\n", total);
#ifdef ACX_SLAB_CLOCK
    {
        unsigned long long ck[4];
        hipMemcpyFromSymbol(ck, HIP_SYMBOL(acx::acx_gs_clock), sizeof(ck));
        printf("in-kernel clock over the k-loops of all launches: %.3f GHz (%.0f cycles, %.2f us per workgroup)\n",
               (double)ck[0] / ((double)ck[1] * 10.0), (double)ck[0] / ck[2], (double)ck[1] / ck[2] / 100.0);
    }
#endif
    return 0;
}
