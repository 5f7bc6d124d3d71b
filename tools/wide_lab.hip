// Stand-alone timing lab for the wide fused MLP kernel (not part of libacx):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DWIDE_SRC='"path/to/mlp_fused_wide.hip"' tools/wide_lab.hip -o /tmp/wide_lab
// tools/lab/build_mlp_labs.sh, build_wide_ablation.sh and build_fused96_ablation.sh build it HERE (the binaries travel to the GPU
// box under build/labs/): the product sources, earlier versions, and sed-patched ablation variants (no DMA, no GELU, ...).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include WIDE_SRC

namespace acx {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
ProfScope::ProfScope(acx_ctx*, int, hipStream_t) : ctx(nullptr) {}
ProfScope::~ProfScope() {}
void prof_next_events(hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
thread_local int tls_inflight_ways = 1;
Tuning& tuning() {     // the library's switches that matter here, straight from the environment
    static Tuning t;
    static bool init = false;
    if (!init) { init = true; if (const char* e = getenv("ACX_WIDE_PERSIST")) t.wide_pers.store(e[0] == '1' ? 1 : 2); }
    return t;
}
}

int main(int argc, char** argv) {
#ifndef WIDE_C
#define WIDE_C 384
#endif
#ifndef WIDE_FN
#define WIDE_FN launch_mlp_fused_wide
#endif
#ifdef WIDE_BF16   /* mlp_fused_wide_bf16.hip: WIDE_BF16 = 0 (fp32 activations in HBM) or 1 (bf16 activations) */
#define WIDE_CALL() acx::launch_mlp_fused_wide_bf16(nullptr, bw, C, y, x, M, 0, nullptr, 0, WIDE_BF16 != 0)
#else
#define WIDE_BF16 0
#define WIDE_CALL() acx::WIDE_FN(nullptr, bw, C, y, x, M, 0)
#endif
    const int C = WIDE_C;
    const long long M = argc > 1 ? atoll(argv[1]) : 64LL * 252 * 56 * 96 / C;
    float *y, *x, *b1, *b2; char* w;
    const size_t wbytes = (size_t)2 * (4 * C / 32) * 128 * C;
    hipMalloc(&y, M * C * 4); hipMalloc(&x, M * C * 4); hipMalloc(&b1, 4 * C * 4); hipMalloc(&b2, C * 4); hipMalloc(&w, wbytes);
    {
        std::vector<float> h((size_t)M * C);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((int)((i * 2654435761u) >> 20 & 0xfff) - 2048) * 1e-3f;
        hipMemcpy(b1, h.data(), 4 * C * 4, hipMemcpyHostToDevice);
        hipMemcpy(b2, h.data(), C * 4, hipMemcpyHostToDevice);
        if (WIDE_BF16) {      // the same values as bf16 (upper halves of the floats), packed
            uint16_t* hb = reinterpret_cast<uint16_t*>(h.data());
            for (size_t i = 0; i < h.size(); ++i) { uint32_t u; std::memcpy(&u, &h[i], 4); hb[i] = (uint16_t)(u >> 16); }
        }
        hipMemcpy(y, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<uint16_t> hw(wbytes / 2);
        for (size_t i = 0; i < hw.size(); ++i) {
            const unsigned r = (unsigned)((i * 2654435761u) >> 9);
            const bool lo = (i >> 3) & 1;
            const _Float16 v = (_Float16)(((int)(r & 0x7ff) - 1024) * (lo ? 1e-4f : 0.01f));
            std::memcpy(&hw[i], &v, 2);
        }
        hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    }
    acx::BlockW bw;
    bw.wstream_s = reinterpret_cast<uint16_t*>(w); bw.wstream_b = reinterpret_cast<uint16_t*>(w); bw.wpack_s = reinterpret_cast<uint16_t*>(w); /* (mlp_fused_split.hip reads wpack_s) */ bw.b1 = b1; bw.b2 = b2; bw.w1s_scale = 1.f; bw.w2s_scale = 1.f; bw.hid_scale = 16.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) if (WIDE_CALL() != 0) return 1;
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) WIDE_CALL();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    const double tf = 4.0 * M * C * 4 * C / (best * 1e-3) / 1e12;
    printf("M=%lld: %.1f us per block, %.1f TF fp32-equivalent = %.3f of 833\n", M, best * 1e3, tf, tf / 833.3);
#ifdef ACX_FW_STAMPS
    {   // wide kernel: per-wave section sums of the LAST launch (s_memtime ticks), averaged over the first 2048 workgroups
        static unsigned long long st[2048 * 4 * 8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(acx::acx_fw_stamps), sizeof(st));
        const char* names[8] = {"prologue", "phase-1 segments", "phase-2 segments", "segment-end wait+barrier", "epilogue", "-", "-", "-"};
        long long wgs = (M + 127) / 128 < 2048 ? (M + 127) / 128 : 2048;
        double per_tile = 1.0;      // persistent form: a workgroup's sums cover all its tiles
        { const char* e = getenv("ACX_WIDE_PERSIST"); int cus = 256; if (C == 192 && !(e && e[0] == '0') && (M + 127) / 128 > cus) { per_tile = (double)((M + 127) / 128) / cus; wgs = cus; } }
        for (int w : {0, 3}) {
            double sum[8] = {0};
            for (int b = 0; b < wgs; ++b) for (int k = 0; k < 8; ++k) sum[k] += (double)st[(b * 4 + w) * 8 + k];
            double tot = 0; for (int k = 0; k < 5; ++k) tot += sum[k];
            printf("   wave %d: ticks per tile:", w);
            for (int k = 0; k < 5; ++k) printf(" %s %.0f |", names[k], sum[k] / wgs / per_tile);
            printf(" total %.0f\n", tot / wgs / per_tile);
        }
    }
#endif
#ifdef ACX_FS_STAMPS
    {   // per-wave section sums of the LAST launch (s_memtime ticks): average over workgroups, waves 0 and 4
        static unsigned long long st[256 * 8 * 8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(acx::acx_fs_stamps), sizeof(st));
        const char* names[8] = {"prologue", "phase1+gelu", "phase2+gelu", "pack+copy", "wait+barrier", "short iters phase2", "epilogue", "-"};
        const long long tiles_total = (M + 255) / 256;
        for (int w : {0, 4}) {
            double sum[8] = {0};
            for (int b = 0; b < 256; ++b) for (int k = 0; k < 8; ++k) sum[k] += (double)st[(b * 8 + w) * 8 + k];
            double tot = 0; for (int k = 0; k < 8; ++k) tot += sum[k];
            printf("   wave %d: ticks per tile:", w);
            for (int k = 0; k < 7; ++k) printf(" %s %.0f |", names[k], sum[k] / tiles_total);
            printf(" total %.0f\n", tot / tiles_total);
        }
    }
#endif
    return 0;
}
