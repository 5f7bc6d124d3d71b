// Lab driver for the matrix-pipe depthwise kernel (csrc/dwconv_mfma.hip) against the column-streaming kernel
// (csrc/dwconv_col.hip) on bf16 activations: results against a host double sum with the SAME bf16-rounded weights (the
// kernels then differ by fp32 summation order and the final bf16 rounding only), and launch times of both.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I audioset-convnext-inf_amd/csrc -I include tools/lab/dwm_lab.hip -o build/dwm_lab
//   build/dwm_lab            product shapes at B = 64 (timed) and odd shapes (checked)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include <vector>
#include <algorithm>

#ifndef DWM_ONLY      // -DDWM_ONLY: the matrix kernel alone, timing only (ablation builds: -DACX_DWM_ABLATE=1..4)
#include "dwconv.hip"
#include "dwconv_col.hip"
#endif
#include "dwconv_mfma.hip"

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
static float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int run_case(int B, int H, int W, int iters, int target_waves, bool check) {
    const int C = 96 * 56 / W;
    const size_t n = (size_t)B * H * W * C;
    std::mt19937 rng(1234 + B + H);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::vector<float> hw(49 * C), hb(C);
    for (auto& v : hw) v = from_bf16(to_bf16(d(rng) * 0.2f));
    for (auto& v : hb) v = d(rng);
    std::vector<uint16_t> hx(n);
    for (size_t i = 0; i < n; ++i) hx[i] = to_bf16(d(rng) * 3.f);
    // the matrix kernel's operand image of the weights (api.hip packs BlockW::dw_ops the same way)
    std::vector<uint16_t> hops((size_t)(C / 32) * 42 * 64 * 4);
    for (int sl = 0; sl < C / 32; ++sl)
        for (int kh = 0; kh < 7; ++kh)
            for (int d3 = 0; d3 < 3; ++d3)
                for (int st = 0; st < 2; ++st)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int q = lane & 3, ch = 32 * sl + 2 * (lane >> 2) + st;
                        for (int k = 0; k < 4; ++k) {
                            const int tp = 4 * d3 + k - q - 1;
                            hops[((((size_t)(sl * 7 + kh) * 3 + d3) * 2 + st) * 64 + lane) * 4 + k] = (tp >= 0 && tp < 7) ? to_bf16(hw[(kh * 7 + tp) * C + ch]) : (uint16_t)0;
                        }
                    }
    void *x, *y0, *y1, *sink, *dops; float *dw, *db;
    CK(hipMalloc(&dops, hops.size() * 2)); CK(hipMemcpy(dops, hops.data(), hops.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y0, n * 2)); CK(hipMalloc(&y1, n * 2)); CK(hipMalloc(&sink, kDwSinkBytes));
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(y0, 0xff, n * 2)); CK(hipMemset(y1, 0xee, n * 2));
    if (launch_dwconv_col(x, y0, dw, db, sink, B, H, W, true, target_waves, nullptr) != ACX_OK) return 1;
    if (launch_dwconv_mfma(x, y1, dops, db, sink, B, H, W, target_waves, nullptr) != ACX_OK) return 1;
    CK(hipDeviceSynchronize());
    int bad = 0;
    if (check) {
        std::vector<uint16_t> a(n), b(n);
        CK(hipMemcpy(a.data(), y0, n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), y1, n * 2, hipMemcpyDeviceToHost));
        size_t diff = 0, far = 0, first = 0, wrong_old = 0, wrong_new = 0;
        for (size_t i = 0; i < n; ++i) {
            if (a[i] == b[i]) continue;
            ++diff;
            const int da = (int)(a[i] & 0x7fff) - (int)(b[i] & 0x7fff);
            if ((a[i] ^ b[i]) & 0x8000 || da > 1 || da < -1) {      // more than one bf16 step apart: compare with the exact sum
                const size_t c = i % C, px = i / C, w = px % W, h = (px / W) % H, bb = px / W / H;
                double s = hb[c];
                for (int ky = 0; ky < 7; ++ky)
                    for (int kx = 0; kx < 7; ++kx) {
                        const int hh = (int)h + ky - 3, ww = (int)w + kx - 3;
                        if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                        s += (double)from_bf16(hx[((bb * H + hh) * W + ww) * C + c]) * hw[(ky * 7 + kx) * C + c];
                    }
                const double tol = std::fabs(s) * (1.0 / 128) + 1e-5;       // a bf16 step around the exact value
                const bool oka = std::fabs(from_bf16(a[i]) - s) <= tol, okb = std::fabs(from_bf16(b[i]) - s) <= tol;
                wrong_old += !oka; wrong_new += !okb;
                if (!okb) { if (!far) first = i; ++far; }
            }
        }
        if (far) {
            bad = 1;
            const size_t c = first % C, px = first / C, w = px % W, h = (px / W) % H, bb = px / W / H;
            printf("  WRONG: %zu of %zu elements off the exact sum; first at clip %zu row %zu col %zu ch %zu: column kernel %g matrix kernel %g\n",
                   far, n, bb, h, w, c, from_bf16(a[first]), from_bf16(b[first]));
        }
        printf("  %zu of %zu elements differ between the kernels by one bf16 step (%.3f %%), column kernel off the exact sum: %zu\n", diff, n, 100.0 * diff / n, wrong_old);
    }
    float t_old = 0, t_new = 0;
    if (iters > 0) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < 5; ++i) launch_dwconv_col(x, y0, dw, db, sink, B, H, W, true, target_waves, nullptr);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) launch_dwconv_col(x, y0, dw, db, sink, B, H, W, true, target_waves, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_old, e0, e1));
            for (int i = 0; i < 5; ++i) launch_dwconv_mfma(x, y1, dops, db, sink, B, H, W, target_waves, nullptr);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) launch_dwconv_mfma(x, y1, dops, db, sink, B, H, W, target_waves, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_new, e0, e1));
        }
    }
    const double mb = 2.0 * n * 2 / 1e6;
    printf("B=%3d H=%3d W=%2d C=%3d bf16 waves=%4d  %s  column %7.1f us (%5.2f TB/s)  matrix %7.1f us (%5.2f TB/s)\n", B, H, W, C,
           target_waves, check ? (bad ? "WRONG" : "ok") : "unchecked", iters ? t_old * 1e3 / iters : 0.0, iters ? mb / (t_old * 1e3 / iters) : 0.0,
           iters ? t_new * 1e3 / iters : 0.0, iters ? mb / (t_new * 1e3 / iters) : 0.0);
    hipFree(x); hipFree(y0); hipFree(y1); hipFree(sink); hipFree(dw); hipFree(db); hipFree(dops);
    return bad;
}

#ifdef DWM_ONLY
namespace acx { int launch_dwconv_col(const void*, void*, const float*, const float*, void*, int, int, int, bool, int, hipStream_t) { return ACX_OK; } }
#endif
int main(int argc, char** argv) {
    int bad = 0;
    const int waves = argc > 1 ? atoi(argv[1]) : 1024;
#ifdef ACX_DWM_STAMPS
    for (int W : {14, 28, 56}) {
        const int B = 64, H = 252 * W / 56, C = 96 * 56 / W; const size_t n = (size_t)B * H * W * C;
        void *x, *y, *sink; float *dw, *db;
        CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&sink, kDwSinkBytes)); CK(hipMalloc(&dw, (C / 32) * 42 * 512)); CK(hipMalloc(&db, C * 4));
        CK(hipMemset(x, 0, n * 2)); CK(hipMemset(dw, 0, (C / 32) * 42 * 512)); CK(hipMemset(db, 0, C * 4));
        std::vector<unsigned long long> st(4096 * 16, 0);
        for (int i = 0; i < 4; ++i) launch_dwconv_mfma(x, y, dw, db, sink, B, H, W, waves, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyToSymbol(HIP_SYMBOL(acx_dwm_stamps), st.data(), st.size() * 8));
        launch_dwconv_mfma(x, y, dw, db, sink, B, H, W, waves, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(acx_dwm_stamps), st.size() * 8));
        unsigned long long t0 = ~0ull, t1 = 0; int items = 0;
        for (int i = 0; i < 4096; ++i) if (st[i * 16 + 15]) { ++items; t0 = std::min(t0, st[i * 16]); t1 = std::max(t1, st[i * 16 + 15]); }
        printf("W=%d: %d waves stamped, first start -> last end %.1f k ticks (100 MHz: 10 ns each)\n  median ticks from the wave's start:", W, items, (t1 - t0) / 1e3);
        const char* names[16] = {"start", "requests out", "weights requested", "rows landed", "first group read", "pair 0", "pair 1", "pair 2", "pair 3", "pair 4", "pair 5", "pair 6", "pair 7", "pair 8", "pair 9", "end"};
        for (int k = 0; k < 16; ++k) {
            std::vector<double> v;
            for (int i = 0; i < 4096; ++i) if (st[i * 16 + 15] && st[i * 16 + k]) v.push_back((double)(st[i * 16 + k] - st[i * 16]));
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  %s %.0f", names[k], v[v.size() / 2]);
        }
        std::vector<double> so;
        for (int i = 0; i < 4096; ++i) if (st[i * 16 + 15]) so.push_back((double)(st[i * 16] - t0));
        std::sort(so.begin(), so.end());
        printf("\n  start offsets: median %.0f p90 %.0f max %.0f\n", so[so.size() / 2], so[so.size() * 9 / 10], so.back());
        hipFree(x); hipFree(y); hipFree(sink); hipFree(dw); hipFree(db);
    }
    return 0;
#endif
#ifdef DWM_PMC
    run_case(64, 63, 14, 2, waves, false); run_case(64, 126, 28, 2, waves, false); run_case(64, 252, 56, 2, waves, false);
    return 0;
#endif
#ifdef DWM_ONLY
    run_case(64, 63, 14, 20, waves, false);
    run_case(64, 126, 28, 20, waves, false);
    run_case(64, 252, 56, 20, waves, false);
    return 0;
#endif
    // odd shapes: one clip, few rows, clip boundaries inside a step, batch ends inside a segment
    bad |= run_case(1, 5, 14, 0, 64, true);
    bad |= run_case(1, 9, 14, 0, 4096, true);      // 12-row segments: an odd number of steps
    bad |= run_case(4, 31, 28, 0, 4096, true);
    bad |= run_case(7, 50, 56, 0, 300, true);
    bad |= run_case(3, 9, 14, 0, 64, true);
    bad |= run_case(2, 31, 28, 0, 64, true);
    bad |= run_case(3, 17, 56, 0, 64, true);
    bad |= run_case(5, 63, 14, 0, 1024, true);
    bad |= run_case(2, 126, 28, 0, 1024, true);
    bad |= run_case(2, 252, 56, 0, 1024, true);
    // product shapes
    bad |= run_case(64, 63, 14, 20, waves, true);
    bad |= run_case(64, 126, 28, 20, waves, true);
    bad |= run_case(64, 252, 56, 20, waves, true);
    for (int w : {512, 2048}) {
        run_case(64, 63, 14, 20, w, false);
        run_case(64, 126, 28, 20, w, false);
        run_case(64, 252, 56, 20, w, false);
    }
    printf(bad ? "FAILED\n" : "all cases ok\n");
    return bad;
}
