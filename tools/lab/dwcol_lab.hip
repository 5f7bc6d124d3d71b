// Lab driver for the column-streaming depthwise kernel (csrc/dwconv_col.hip) against the ring / tile kernels of
// csrc/dwconv.hip: bit-for-bit comparison on the product shapes and on odd ones, and launch times of both.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I audioset-convnext-inf_amd/csrc tools/lab/dwcol_lab.hip -o build/dwcol_lab
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>

#include "dwconv.hip"
#include "dwconv_col.hip"

namespace acx {
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}
thread_local int tls_inflight_ways = 1;
Tuning& tuning() { static Tuning t; return t; }
ProfScope::ProfScope(acx_ctx*, int k, hipStream_t st) : ctx(nullptr), cls(k), s(st), prev(nullptr) {}
ProfScope::~ProfScope() {}
void prof_next_events(hipEvent_t*, hipEvent_t*) {}
}  // namespace acx
using namespace acx;

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

int run_case(int B, int H, int W, bool bf, int iters, int target_waves, bool check) {
    const int C = 96 * 56 / W;
    const size_t n = (size_t)B * H * W * C, esz = bf ? 2 : 4;
    std::mt19937 rng(1234 + B + H);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::vector<float> hw(49 * C), hb(C);
    for (auto& v : hw) v = d(rng) * 0.2f;
    for (auto& v : hb) v = d(rng);
    std::vector<char> hx(n * esz);
    for (size_t i = 0; i < n; ++i) {
        float v = d(rng) * 3.f;
        if (bf) { uint16_t q = to_bf16(v); memcpy(&hx[i * 2], &q, 2); } else memcpy(&hx[i * 4], &v, 4);
    }
    void *x, *y0, *y1, *sink; float *dw, *db;
    CK(hipMalloc(&x, n * esz)); CK(hipMalloc(&y0, n * esz)); CK(hipMalloc(&y1, n * esz)); CK(hipMalloc(&sink, kDwSinkBytes));
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMemcpy(x, hx.data(), n * esz, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(y0, 0xff, n * esz)); CK(hipMemset(y1, 0xee, n * esz));
    BlockW bw; bw.dw = dw; bw.dwb = db;
    if (launch_dwconv(nullptr, bw, C, x, y0, nullptr, B, H, W, nullptr, bf) != ACX_OK) return 1;
    if (launch_dwconv_col(x, y1, dw, db, sink, B, H, W, bf, target_waves, nullptr) != ACX_OK) return 1;
    CK(hipDeviceSynchronize());
    int bad = 0;
    if (check) {
        std::vector<char> a(n * esz), b(n * esz);
        CK(hipMemcpy(a.data(), y0, n * esz, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), y1, n * esz, hipMemcpyDeviceToHost));
        size_t diff = 0, first = 0;
        for (size_t i = 0; i < n * esz; i += esz)
            if (memcmp(&a[i], &b[i], esz) != 0) { if (!diff) first = i / esz; ++diff; }
        if (diff) {
            bad = 1;
            const size_t c = first % C, px = first / C, w = px % W, h = (px / W) % H, bb = px / W / H;
            float va = 0, vb = 0;
            if (!bf) { memcpy(&va, &a[first * 4], 4); memcpy(&vb, &b[first * 4], 4); }
            printf("  MISMATCH: %zu of %zu elements differ; first at clip %zu row %zu col %zu ch %zu: old %g new %g\n", diff, n, bb, h, w, c, va, vb);
        }
    }
    float t_old = 0, t_new = 0;
    if (iters > 0) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 5; ++i) launch_dwconv(nullptr, bw, C, x, y0, nullptr, B, H, W, nullptr, bf);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) launch_dwconv(nullptr, bw, C, x, y0, nullptr, B, H, W, nullptr, bf);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_old, e0, e1));
            for (int i = 0; i < 5; ++i) launch_dwconv_col(x, y1, dw, db, sink, B, H, W, bf, target_waves, nullptr);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) launch_dwconv_col(x, y1, dw, db, sink, B, H, W, bf, target_waves, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_new, e0, e1));
        }
    }
    const double mb = 2.0 * n * esz / 1e6;
    printf("B=%3d H=%3d W=%2d C=%3d %s waves=%4d  %s  old %7.1f us (%5.2f TB/s)  new %7.1f us (%5.2f TB/s)\n", B, H, W, C, bf ? "bf16" : "fp32",
           target_waves, check ? (bad ? "DIFF" : "same bits") : "unchecked", iters ? t_old * 1e3 / iters : 0.0, iters ? mb / (t_old * 1e3 / iters) : 0.0,
           iters ? t_new * 1e3 / iters : 0.0, iters ? mb / (t_new * 1e3 / iters) : 0.0);
    hipFree(x); hipFree(y0); hipFree(y1); hipFree(sink); hipFree(dw); hipFree(db);
    return bad;
}

#ifdef ACX_DWC_STAMPS
static void stamp_case(int B, int H, int W, int waves) {
    const int C = 96 * 56 / W; const size_t n = (size_t)B * H * W * C;
    void *x, *y, *sink; float *dw, *db;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&sink, kDwSinkBytes)); CK(hipMalloc(&dw, 49 * C * 4)); CK(hipMalloc(&db, C * 4));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(dw, 0, 49 * C * 4)); CK(hipMemset(db, 0, C * 4));
    std::vector<unsigned long long> st(4096 * 8, 0);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(acx_dwc_stamps), st.data(), st.size() * 8));
    for (int i = 0; i < 6; ++i) launch_dwconv_col(x, y, dw, db, sink, B, H, W, false, waves, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(acx_dwc_stamps), st.size() * 8));
    unsigned long long t0 = ~0ull, t6 = 0; int items = 0;
    for (int i = 0; i < 4096; ++i) if (st[i * 8 + 6]) { ++items; if (st[i * 8] < t0) t0 = st[i * 8]; if (st[i * 8 + 6] > t6) t6 = st[i * 8 + 6]; }
    double avg[7] = {0};
    CK(hipMemset(x, 0, 64));
    for (int i = 0; i < 4096; ++i) if (st[i * 8 + 6]) for (int k = 0; k < 7; ++k) avg[k] += (double)(st[i * 8 + k] - st[i * 8]) / items;
    {
        std::vector<double> life, startoff;
        for (int i = 0; i < 4096; ++i) if (st[i * 8 + 6]) { life.push_back((double)(st[i * 8 + 6] - st[i * 8])); startoff.push_back((double)(st[i * 8] - t0)); }
        std::sort(life.begin(), life.end()); std::sort(startoff.begin(), startoff.end());
        printf("W=%d: lifetime k cycles min %.1f median %.1f p90 %.1f max %.1f | start offset median %.1f p90 %.1f max %.1f | first start -> last end %.1f\n", W,
               life.front() / 1e3, life[life.size() / 2] / 1e3, life[life.size() * 9 / 10] / 1e3, life.back() / 1e3,
               startoff[startoff.size() / 2] / 1e3, startoff[startoff.size() * 9 / 10] / 1e3, startoff.back() / 1e3, (t6 - t0) / 1e3);
    }
    printf("W=%d B=%d: %d waves; mean stamp (k cycles from the wave's own start):", W, B, items);
    const char* names[7] = {"start", "prologue issued", "first rows landed", "group 0 done", "main loop done", "epilogue done", "stores drained"};
    for (int k = 0; k < 7; ++k) printf("  %s %.1f", names[k], avg[k] / 1e3);
    printf("\n");
    hipFree(x); hipFree(y); hipFree(sink); hipFree(dw); hipFree(db);
}
#endif

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 1024;
    int bad = 0;
    const int Hs[4] = {252, 126, 63, 31}, Ws[4] = {56, 28, 14, 7};
#ifdef ACX_DWC_STAMPS
    for (int s = 0; s < 4; ++s) stamp_case(64, Hs[s], Ws[s], waves);
    return 0;
#endif
    if (argc > 2) {          // timing only: B=64 at `waves`, B=32 at waves and waves/2
        for (int s = 0; s < 4; ++s) run_case(64, Hs[s], Ws[s], false, 20, waves, false);
        for (int s = 0; s < 4; ++s) run_case(32, Hs[s], Ws[s], false, 20, waves, false);
        for (int s = 0; s < 3; ++s) run_case(64, Hs[s], Ws[s], true, 20, waves, false);
        return 0;
    }
    // odd shapes first (edges: 1 clip, few rows, rows fewer than the halo, many short clips)
    for (int s = 0; s < 4; ++s)
        for (int B : {1, 2, 5})
            for (int H : {1, 2, 3, 7, 23})
                bad += run_case(B, H >> (s > 1 ? 0 : 0), Ws[s], false, 0, waves, true);
    for (int s = 0; s < 3; ++s) bad += run_case(3, 9, Ws[s], true, 0, waves, true);
    for (int s = 0; s < 4; ++s) bad += run_case(4, Hs[s], Ws[s], false, 0, waves, true);
    // product shapes, timed
    for (int B : {64, 32})
        for (int s = 0; s < 4; ++s) bad += run_case(B, Hs[s], Ws[s], false, 20, waves / (B == 32 ? 2 : 1), true);
    for (int s = 0; s < 3; ++s) bad += run_case(64, Hs[s], Ws[s], true, 20, waves, true);
    for (int s = 0; s < 4; ++s) bad += run_case(64, Hs[s], Ws[s], false, 20, 2 * waves, false);
    printf(bad ? "FAILED (%d cases differ)\n" : "all cases identical\n", bad);
    return bad != 0;
}
