// Stand-alone timing lab for the fp32-MFMA GEMM kernel (diagnostic builds; not part of libacx).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DACX_LAB_xxx] tools/gemm_lab.hip -o build/gemm_lab
//   build/gemm_lab          -> TFLOP/s for the backbone's GEMM shapes at batch 64
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../audioset-convnext-inf_amd/csrc/gemm.hip"

namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
}

int main(int argc, char** argv) {
    struct Shape { const char* name; long long M; int N, K; int epi; bool ln; };
    const long long P0 = 64LL * 252 * 56, P1 = P0 / 4, P2 = P1 / 4, P3 = 64LL * 31 * 7;
    std::vector<Shape> shapes = {
        {"s0.pw1", P0, 384, 96, acx::EPI_GELU, true},   {"s0.pw2", P0, 96, 384, acx::EPI_RESID, false},
        {"s1.pw1", P1, 768, 192, acx::EPI_GELU, true},  {"s1.pw2", P1, 192, 768, acx::EPI_RESID, false},
        {"s2.pw1", P2, 1536, 384, acx::EPI_GELU, true}, {"s2.pw2", P2, 384, 1536, acx::EPI_RESID, false},
        {"s3.pw1", P3, 3072, 768, acx::EPI_GELU, true}, {"s3.pw2", P3, 768, 3072, acx::EPI_RESID, false},
    };
    size_t maxA = 0, maxO = 0;
    for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxO = std::max(maxO, (size_t)s.M * s.N); }
    float *A, *W, *O, *bias, *stats;
    hipMalloc(&A, maxA * 4); hipMalloc(&O, maxO * 4); hipMalloc(&W, (size_t)3072 * 768 * 4);
    hipMalloc(&bias, 3072 * 4); hipMalloc(&stats, (size_t)P0 * 8);
    {
        std::vector<float> h(maxA);
        for (size_t i = 0; i < maxA; ++i) h[i] = (float)((int)((i * 2654435761u) >> 8 & 0xffff) - 32768) / 32768.f;
        hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice);
        hipMemcpy(O, h.data(), std::min(maxA, maxO) * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)3072 * 768 * 4, hipMemcpyHostToDevice);
        hipMemcpy(bias, h.data(), 3072 * 4, hipMemcpyHostToDevice);
        std::vector<float> st((size_t)P0 * 2);
        for (size_t i = 0; i < (size_t)P0; ++i) { st[2 * i] = 0.01f; st[2 * i + 1] = 1.1f; }
        hipMemcpy(stats, st.data(), st.size() * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 5;
    for (auto& s : shapes) {
        acx::GemmArgs g{};
        g.A = A; g.Wt = W; g.bias = bias; g.out = O; g.stats = s.ln ? stats : nullptr; g.colsum = s.ln ? bias : nullptr; g.resid = s.epi == acx::EPI_RESID ? O : nullptr;
        g.M = s.M; g.N = s.N; g.K = s.K; g.epi = s.epi; g.cls = 0;
        if (acx::launch_gemm(nullptr, g, 0) != 0) return 1;
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) acx::launch_gemm(nullptr, g, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        double tf = 2.0 * s.M * s.N * s.K / (ms * 1e-3) / 1e12;
        printf("%-8s M=%-8lld N=%-5d K=%-5d  %8.1f us  %6.1f TF  (%4.1f%% of 157.3)\n", s.name, s.M, s.N, s.K, ms * 1e3, tf, 100 * tf / 157.3);
#ifdef ACX_LAB_GEMM_STAMP
        unsigned long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(acx::acx_gemm_stamps), sizeof(st));
        double nt = (double)st[5], nw = (double)st[6];
        printf("   per k-tile cycles: g0-g2 (48 MFMA + DMA + reads) %.0f | barrier %.0f | g3 (16 MFMA + reads) %.0f  || per wave: main loop %.0f, last tile %.0f, epilogue %.0f (k-tiles/wave %.1f)\n",
               st[0] / nt, st[1] / nt, st[2] / nt, st[3] / nw, st[7] / nw, st[4] / nw, nt / nw);
        unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(acx::acx_gemm_stamps), z, sizeof(z));
#endif
    }
    return 0;
}
