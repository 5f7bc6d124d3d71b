// EXPERIMENT, not built into libacx (kept as the record of a measured negative): persistent, wave-specialised split GEMM.
// Correct (the whole GPU parity suite passed with it in place of gemm_split for pwconv1/pwconv2 of stages 2-3) but
// SLOWER than gemm_split.hip on MI355X: s2.pw1 332 vs 291 us, s2.pw2 263 vs 240, s3.pw1 253 vs 212, s3.pw2 229 vs 228.
// Without operand traffic its four consumer waves per CU nearly match the eight mixed waves of gemm_split (180 vs 161
// us); with it they lose 90 us (vs 57): four loader waves issuing 8 LDS-DMA pieces each per step (60-180 cycles of
// issue apiece) cannot feed a step of 768 MFMA cycles, and the per-step barrier ties the consumers to them.  More
// loader waves (-DACX_WS_LOADERS=8: 12 waves, 168 VGPRs, the epilogue spills) change nothing in the main loop either
// (260 vs 271 us without epilogue): the loaders wait for data, not for issue slots.  What would help: fewer operand
// bytes per flop (256-row tiles).  To try it again: add it to csrc/Makefile, declare the two entry points in
// acx_internal.h and route run_mlp_split through launch_gemm_split_ws when gemm_split_ws_supported().
// K4w -- split-fp16 GEMM (see gemm_split.hip for the arithmetic and the S16 operand format) as a PERSISTENT,
// WAVE-SPECIALISED kernel for the large pointwise contractions of stages 2-3:
//
//   one workgroup of 8 waves per CU walks over its 128 x 128 output tiles;
//   waves 4-7 (loaders)   only issue LDS-DMA: the k-tiles of all the workgroup's tiles form ONE stream of "steps"
//                         that they keep kStages-1 steps ahead in a ring of kStages stages (A 16 KB + B 16 KB each),
//                         across tile boundaries, behind a counted vmcnt;
//   waves 0-3 (consumers) only read fragments and issue MFMAs (64 x 64 per wave), then run the tile's epilogue while
//                         the loaders are already fetching the next tile.
// Why: in gemm_split.hip every wave pays 60-180 cycles of issue per LDS-DMA piece inside a k-tile of only 768 MFMA
// cycles, drains its loads at every barrier, and -- with an LDS-DMA in its own instruction stream -- gets only
// lgkmcnt(0) waits from hipcc; prologue latency and epilogue are exposed once per 13-us workgroup.  Here the consumer
// waves' stream contains neither loads nor stores until the epilogue.
//
// Synchronisation: ONE s_barrier per step, executed by all 8 waves.  Step g lives in stage g % kStages.
//   loaders,   iteration g: wait until step g has landed (vmcnt leaves the kStages-2 younger steps in flight) | barrier g |
//                           issue step g + kStages - 1 into the stage of step g - 1 (its readers finished before barrier g)
//   consumers, iteration g: barrier g | issue the fragment reads of step g | MFMAs of step g-1 (registers) |
//                           [tile finished: epilogue] | wait for the reads (they must be complete before barrier g+1
//                           lets the loaders overwrite that stage)
#include "../../audioset-convnext-inf_amd/csrc/acx_internal.h"
#include "../../audioset-convnext-inf_amd/csrc/split_math.h"

namespace acx {

constexpr int kWsStages = 4;
constexpr int kWsRowBytes = 128;
constexpr int kWsBK = 32;
constexpr int kWsTile = 128 * kWsRowBytes;          // one operand tile of one step: 128 rows x 128 B = 16 KB
constexpr size_t kWsLdsBytes = (size_t)kWsStages * 2 * kWsTile;

struct GemmWsParams {
    const char* A; const char* Wt; const float* bias; void* out; const float* resid;
    long long M; int N; int K;
    float sinv;
    int tiles_n; long long tiles;         // 128 x 128 tiles, n fastest
};

__device__ __forceinline__ void lds_dma16_w(const char* gsrc, char* lds_wave_base) {
#ifdef ACX_SLAB_NO_DMA      // diagnostic (tools/split_lab.hip)
    return;
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// tile handled by workgroup `bid` in its i-th round: rounds are contiguous runs of gridDim.x tiles, and inside a round
// the XCD-contiguous remap of gemm.hip keeps the tiles that share an A row panel on one XCD
__device__ __forceinline__ long long ws_tile_of(long long i, long long bid, long long nwg, long long tiles) {
    const long long base = i * nwg;
    long long cnt = tiles - base;
    if (cnt > nwg) cnt = nwg;
    if (bid >= cnt) return -1;
    const long long per = (cnt + 7) >> 3, full = cnt - (per - 1) * 8;
    const long long xcd = bid & 7, k = bid >> 3;
    return base + (xcd < full ? xcd * per : full * per + (xcd - full) * (per - 1)) + k;
}

// EPI: 1 bias + GELU -> S16 (scaled by kSplitHiddenScale), 2 bias + residual -> fp32
#ifndef ACX_WS_LOADERS
#define ACX_WS_LOADERS 4
#endif
constexpr int kWsLoaders = ACX_WS_LOADERS;          // loader waves (4 or 8); each stages 128 / kWsLoaders rows of A and of B
constexpr int kWsLP = 16 / kWsLoaders;              // 1-KB pieces per operand per loader per step
template <int EPI>
__global__ __launch_bounds__(256 + 64 * kWsLoaders) void gemm_split_ws_kernel(GemmWsParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [kWsStages][A tile | B tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / kWsBK;
    const long long nwg = gridDim.x, bid = blockIdx.x;
    // number of tiles of this workgroup (every wave computes the same)
    long long my_tiles = 0;
    for (long long i = 0; ws_tile_of(i, bid, nwg, p.tiles) >= 0; ++i) ++my_tiles;
    const long long total_steps = my_tiles * nk;

    if (wave >= 4) {
        // =========================================== loaders ===========================================================
        const int lw = wave - 4;
        const int prow = lane >> 3, pchunk = lane & 7;
        // per step this wave stages rows [8 kWsLP lw, +8 kWsLP) of A and of B: kWsLP + kWsLP pieces of 8 rows
        const char* a_src[kWsLP];
        const char* b_src[kWsLP];
        char* const dst0 = smem + (8 * kWsLP * lw) * kWsRowBytes;
        long long cur_i = -1;
        auto set_tile = [&](long long i) {
            const long long t = ws_tile_of(i, bid, nwg, p.tiles);
            const long long tile_m = t / p.tiles_n;
            const int tile_n = (int)(t - tile_m * p.tiles_n);
#pragma unroll
            for (int q = 0; q < kWsLP; ++q) {
                const int row = 8 * kWsLP * lw + 8 * q + prow;
                const int chunk = pchunk ^ ((row >> 1) & 7);
                long long m = tile_m * 128 + row;
                if (m >= p.M) m = p.M - 1;
                a_src[q] = p.A + m * p.K * 4 + 16 * chunk;
                b_src[q] = p.Wt + (long long)(tile_n * 128 + row) * p.K * 4 + 16 * chunk;
            }
            cur_i = i;
        };
        auto issue = [&](long long g) {          // step g = (tile round g / nk, k-tile g % nk) -> stage g % kWsStages
            const long long i = g / nk;
            const int kt = (int)(g - i * nk);
            if (i != cur_i) set_tile(i);
            char* d = dst0 + (int)(g % kWsStages) * (2 * kWsTile);
            const long long ko = (long long)kt * kWsBK * 4;
#pragma unroll
            for (int q = 0; q < kWsLP; ++q) lds_dma16_w(a_src[q] + ko, d + q * 8 * kWsRowBytes);
#pragma unroll
            for (int q = 0; q < kWsLP; ++q) lds_dma16_w(b_src[q] + ko, d + kWsTile + q * 8 * kWsRowBytes);
        };
        for (long long g = 0; g < kWsStages - 1 && g < total_steps; ++g) issue(g);
        for (long long g = 0; g <= total_steps; ++g) {
            // step g landed?  (8 pieces per step and wave; the kWsStages-2 younger steps may stay in flight)
            if (g + kWsStages - 2 < total_steps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kWsStages - 2) * 2 * kWsLP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (g + kWsStages - 1 < total_steps) issue(g + kWsStages - 1);
        }
        return;
    }

    // ============================================= consumers ===========================================================
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 consumer waves, 64 x 64 each: TM = TN = 2
    f32x16 acc[2][2];
    const int sw = (l31 >> 1) & 7;
    int foff_hi[2], foff_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        foff_hi[s] = l31 * kWsRowBytes + (((4 * s + 2 * hh) ^ sw) << 4);
        foff_lo[s] = l31 * kWsRowBytes + (((4 * s + 2 * hh + 1) ^ sw) << 4);
    }
    const int a_frag_off = wm * 64 * kWsRowBytes;
    const int b_frag_off = kWsTile + wn * 64 * kWsRowBytes;
    // fragments of ONE k-step (16 k) of a step: F0 = k-step 0, F1 = k-step 1; hi / lo halves of 2 A and 2 B tiles
    f32x4 F0ah[2], F0al[2], F0bh[2], F0bl[2], F1ah[2], F1al[2], F1bh[2], F1bl[2];
#define ACX_WS_READ(F, stage_, s_)                                                                             \
    {                                                                                                          \
        const char* sb_ = smem + (stage_) * (2 * kWsTile);                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
            F##ah[i] = *reinterpret_cast<const f32x4*>(sb_ + a_frag_off + i * 32 * kWsRowBytes + foff_hi[s_]); \
            F##al[i] = *reinterpret_cast<const f32x4*>(sb_ + a_frag_off + i * 32 * kWsRowBytes + foff_lo[s_]); \
            F##bh[i] = *reinterpret_cast<const f32x4*>(sb_ + b_frag_off + i * 32 * kWsRowBytes + foff_hi[s_]); \
            F##bl[i] = *reinterpret_cast<const f32x4*>(sb_ + b_frag_off + i * 32 * kWsRowBytes + foff_lo[s_]); \
        }                                                                                                      \
    }
#define ACX_WS_H8(x) __builtin_bit_cast(h8, x)
    // D = W A^T (lane = row m, registers = 4 consecutive n), terms lo x hi, hi x lo, hi x hi, term-major
#define ACX_WS_MFMA(F)                                                                                         \
    {                                                                                                          \
        _Pragma("unroll") for (int term = 0; term < 3; ++term)                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_WS_H8(term == 0 ? F##bl[j] : F##bh[j]),     \
                                                               ACX_WS_H8(term == 1 ? F##al[i] : F##ah[i]), acc[i][j], 0, 0, 0); \
    }
#define ACX_WS_TOUCH(F)                                                                                        \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
            asm volatile("" :: "v"(F##ah[i]), "v"(F##al[i]), "v"(F##bh[i]), "v"(F##bl[i]));                    \
    }
    const float sinv = p.sinv;
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto epilogue = [&](long long t) {
#ifdef ACX_SLAB_NO_EPI      // diagnostic: keep the accumulators alive, write nothing
        {
            float z = 0.f;
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)
                _Pragma("unroll") for (int r = 0; r < 16; ++r) z += acc[i][j][r];
            if (z == 12345.678f) reinterpret_cast<float*>(p.out)[tid] = z;
            return;
        }
#endif
        const long long tile_m = t / p.tiles_n;
        const int tile_n = (int)(t - tile_m * p.tiles_n);
        const long long m0 = tile_m * 128;
        const int n0 = tile_n * 128;
        // every load of the epilogue is issued before its first store: bias (and out) may alias for all hipcc knows,
        // so a load placed after a store is neither hoisted nor overlapped -- and its vmcnt wait is also a wait for
        // every older store
        f32x4 bq[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[j][q] = *reinterpret_cast<const f32x4*>(p.bias + n0 + (wn * 2 + j) * 32 + 8 * q + 4 * hh);
        if (EPI == 1) {
            GeluConsts gk;
            gk.ps = 0.3275911f * 0.70710678f * sinv;
            gk.cq = 0.84932180f * sinv;
            gk.ca = -0.5f * sinv * kSplitHiddenScale;
            gk.cb = sinv * kSplitHiddenScale;
            const float binv = 1.0f / sinv;
            char* outb = reinterpret_cast<char*>(p.out);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long long m = m0 + (wm * 2 + i) * 32 + l31;
                const bool ok = m < p.M;
                char* orow = outb + (ok ? m : 0) * p.N * 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nb = n0 + (wn * 2 + j) * 32;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 b4 = bq[j][q];
                        unsigned xh[2], xl[2];
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            f32x2 a2, av, tt, ex, g;
                            a2.x = acc[i][j][4 * q + 2 * e2] + b4[2 * e2] * binv;
                            a2.y = acc[i][j][4 * q + 2 * e2 + 1] + b4[2 * e2 + 1] * binv;
                            gelu_piece1(a2, gk, av, tt, ex);
                            gelu_piece2(a2, av, tt, ex, gk, g);
                            gelu_piece3(g, xh[e2], xl[e2]);
                        }
                        auto r0 = __builtin_amdgcn_permlane32_swap(xh[0], xl[0], false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(xh[1], xl[1], false, false);
                        uint4 o;
                        o.x = r0[0]; o.y = r1[0]; o.z = r0[1]; o.w = r1[1];
                        if (ok) *reinterpret_cast<uint4*>(orow + (long long)(nb + 8 * q) * 4 + 16 * hh) = o;
                    }
                }
            }
        } else {
            float* outf = reinterpret_cast<float*>(p.out);
            f32x4 rvv[2][2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long long m = m0 + (wm * 2 + i) * 32 + l31;
                const long long row = (m < p.M ? m : 0) * p.N;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        rvv[i][j][q] = *reinterpret_cast<const f32x4*>(p.resid + row + n0 + (wn * 2 + j) * 32 + 8 * q + 4 * hh);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long long m = m0 + (wm * 2 + i) * 32 + l31;
                const bool ok = m < p.M;
                const long long row = (ok ? m : 0) * p.N;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nb = n0 + (wn * 2 + j) * 32;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 b4 = bq[j][q];
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[i][j][4 * q + e], sinv, b4[e]) + rvv[i][j][q][e];
                        if (ok) *reinterpret_cast<f32x4*>(outf + row + nb + 8 * q + 4 * hh) = v;
                    }
                }
            }
        }
    };

    zero_acc();
    // iteration g:  barrier g | read F0 <- (step g, k-step 0) | MFMAs on F1 = (step g-1, k-step 1) | epilogue if that
    //               closed a tile (after the read of F1 <- (step g, k-step 1) went out) | MFMAs on F0 | reads complete (before barrier g+1
    //               lets the loaders overwrite the stage).  First and last iteration peeled: no conditional around MFMAs.
    long long g = 1;
    int kt_prev = 0;                     // k-tile index of step g-1 inside its tile
    long long i_prev = 0;                // tile round of step g-1
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ACX_WS_READ(F0, 0, 0)
    ACX_WS_READ(F1, 0, 1)
    __builtin_amdgcn_sched_barrier(0);
    ACX_WS_TOUCH(F0)
    ACX_WS_MFMA(F0)
    __builtin_amdgcn_sched_barrier(0);
    ACX_WS_TOUCH(F1)
    for (; g < total_steps; ++g) {
        const int stage = (int)(g % kWsStages);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ACX_WS_READ(F0, stage, 0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_WS_MFMA(F1)
        __builtin_amdgcn_sched_barrier(0);
        // (the second read goes out BEFORE a possible epilogue: issued after it, it would have to wait for the
        //  epilogue's stores -- hipcc reuses F1's registers for store data and then protects them with vmcnt(0))
        ACX_WS_READ(F1, stage, 1)
        __builtin_amdgcn_sched_barrier(0);
        if (kt_prev == nk - 1) {
            epilogue(ws_tile_of(i_prev, bid, nwg, p.tiles));
            zero_acc();
            kt_prev = -1; ++i_prev;
        }
        __builtin_amdgcn_sched_barrier(0);
        ACX_WS_TOUCH(F0)
        ACX_WS_MFMA(F0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_WS_TOUCH(F1)
        ++kt_prev;
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ACX_WS_MFMA(F1)
    epilogue(ws_tile_of(i_prev, bid, nwg, p.tiles));
#undef ACX_WS_READ
#undef ACX_WS_MFMA
#undef ACX_WS_TOUCH
#undef ACX_WS_H8
}

template <int EPI>
static int launch_ws_epi(const GemmWsParams& p, hipStream_t s) {
    static bool attr_set = false;
    static int num_cu = 256;
    if (!attr_set) {
        ACX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_ws_kernel<EPI>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWsLdsBytes));
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const long long blocks = p.tiles < num_cu ? p.tiles : num_cu;
    gemm_split_ws_kernel<EPI><<<dim3((unsigned)blocks), dim3(256 + 64 * kWsLoaders), kWsLdsBytes, s>>>(p);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

bool gemm_split_ws_supported(const GemmSplitArgs& a) {
    return !a.gather && (a.epi == EPI_GELU || a.epi == EPI_RESID) && a.N % 128 == 0 && a.K % kWsBK == 0 &&
           a.K >= 4 * kWsBK && a.M >= 128 * 64;
}

int launch_gemm_split_ws(acx_ctx* c, const GemmSplitArgs& a, hipStream_t s) {
    if (!gemm_split_ws_supported(a)) ACX_FAIL(ACX_ERR_ARG, "gemm_split_ws: unsupported shape");
    GemmWsParams p;
    p.A = reinterpret_cast<const char*>(a.A); p.Wt = reinterpret_cast<const char*>(a.Wt); p.bias = a.bias;
    p.out = a.out; p.resid = a.resid; p.M = a.M; p.N = a.N; p.K = a.K; p.sinv = a.sinv;
    p.tiles_n = a.N / 128;
    p.tiles = ((a.M + 127) / 128) * p.tiles_n;
    ProfScope ps(c, a.cls, s);
    return a.epi == EPI_GELU ? launch_ws_epi<1>(p, s) : launch_ws_epi<2>(p, s);
}

}  // namespace acx
