// Stand-alone timing lab for gemm_split_kernel (not part of libacx):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DGEMM_SRC='"path/to/gemm_split.hip"' tools/gemm_split_lab.hip -o /tmp/gemm_split_lab
// tools/run_gemm_split_lab.sh builds sed-patched variants of the product source (no DMA, no fragment reads, ...) and runs them.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include GEMM_SRC

namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
}

static void run(const char* name, long long M, int N, int K, int epi) {
    char *A, *W; float *bias, *resid; char* out;
    hipMalloc(&A, M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&bias, N * 4); hipMalloc(&resid, M * N * 4); hipMalloc(&out, M * N * 4);
    {
        std::vector<uint16_t> h((size_t)M * K * 2);
        for (size_t i = 0; i < h.size(); ++i) {
            const unsigned r = (unsigned)((i * 2654435761u) >> 9);
            const bool lo = (i >> 3) & 1;
            const _Float16 v = (_Float16)(((int)(r & 0x7ff) - 1024) * (lo ? 1e-4f : 0.01f));
            std::memcpy(&h[i], &v, 2);
        }
        hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
        hipMemset(bias, 0, N * 4); hipMemset(resid, 0, M * N * 4);
    }
    acx::GemmSplitArgs a{};
    a.A = A; a.Wt = W; a.bias = bias; a.out = out; a.resid = resid; a.M = M; a.N = N; a.K = K; a.sinv = 1.f / 1024; a.hscale = 16.f;
    a.gather = 0; a.epi = epi; a.cls = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) if (acx::launch_gemm_split(nullptr, a, 0) != 0) exit(1);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) acx::launch_gemm_split(nullptr, a, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    const double tf = 2.0 * M * N * K / (best * 1e-3) / 1e12;
    printf("%-14s M=%lld N=%d K=%d: %.1f us, %.1f TF fp32-equivalent = %.3f of 833\n", name, M, N, K, best * 1e3, tf, tf / 833.3);
    hipFree(A); hipFree(W); hipFree(bias); hipFree(resid); hipFree(out);
}

int main() {
    run("s3.pw1", 13888, 3072, 768, acx::EPI_GELU);
    run("s3.pw2", 13888, 768, 3072, acx::EPI_RESID);
    run("full4rounds", 65536, 768, 3072, acx::EPI_RESID);      // 256 x 4 workgroups = 4 full rounds, long K
    run("full4r.shortK", 65536, 768, 384, acx::EPI_RESID);
    return 0;
}
