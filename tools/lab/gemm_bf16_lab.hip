// Stand-alone timing lab for gemm_bf16_kernel (not part of libacx): the stage-3 pointwise GEMMs of the bf16 arithmetics at B = 64.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DGEMM_SRC='"path/to/gemm_bf16.hip"' tools/lab/gemm_bf16_lab.hip -o build/labs/gemm_bf16_lab
// tools/lab/build_gemm_bf16_ablation.sh builds the product source and sed-patched ablation variants (outputs wrong by construction).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include GEMM_SRC

namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
void prof_next_events(hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
thread_local int tls_inflight_ways = 1;
Tuning& tuning() { static Tuning t; return t; }
}

static double run(const char* name, long long M, int N, int K, int epi) {
    void *A, *W, *out; float *bias, *resid;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&bias, N * 4);
    hipMalloc(&out, (size_t)M * N * 4); hipMalloc(&resid, (size_t)M * N * 4);
    {
        std::vector<uint16_t> h((size_t)M * K);
        for (size_t i = 0; i < h.size(); ++i) { float v = (float)((int)((i * 2654435761u) >> 20 & 0xfff) - 2048) * 1e-3f; uint32_t u; memcpy(&u, &v, 4); h[i] = (uint16_t)(u >> 16); }
        hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)N * K * 2 < h.size() * 2 ? (size_t)N * K * 2 : h.size() * 2, hipMemcpyHostToDevice);
        hipMemset(bias, 0, N * 4); hipMemset(resid, 0, (size_t)M * N * 4);
    }
    acx::GemmBf16Args a{};
    a.A = A; a.Wt = W; a.bias = bias; a.out = out; a.resid = resid; a.M = M; a.N = N; a.Kp = K; a.lda = K;
    a.gather = 0; a.epi = epi; a.cls = 0; a.out_bf16 = 0;
    for (int i = 0; i < 3; ++i) if (acx::launch_gemm_bf16(nullptr, a, 0) != 0) { printf("launch failed\n"); return 0; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) acx::launch_gemm_bf16(nullptr, a, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    const double tf = 2.0 * M * N * K / (best * 1e-3) / 1e12;
    printf("%-8s M=%lld N=%d K=%d: %.1f us, %.0f TFLOP/s = %.3f of 2500\n", name, M, N, K, best * 1e3, tf, tf / 2500.0);
    hipFree(A); hipFree(W); hipFree(bias); hipFree(out); hipFree(resid);
    return best;
}

int main(int argc, char** argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 14336;      // 64 clips x 32 x 7 pixels of stage 3
    run("pwconv1", M, 3072, 768, acx::EPI_GELU);
    run("pwconv2", M, 768, 3072, acx::EPI_RESID);
    return 0;
}
