// K3 -- depthwise 7x7 convolution, zero pad 3, + bias, channels-last (Block.dwconv,
// convnext.py:58-60 called at :76), and the per-pixel LayerNorm statistics that Block.norm
// (convnext.py:61,78; F.layer_norm :532-535) needs.
//
// HBM-bound: algorithmic bytes = read x + write y = 2*C*H*W*4 B per clip per block
// (10.84 / 5.42 / 2.71 / 1.33 MB in stages 0-3).  12.25 FLOP/B, so the VALU must stay under
// ~50 % busy to reach the HBM roofline: lanes run along C (float4 = 16 B per lane, 8 lanes = one
// 128-B line per pixel), each thread keeps WT=7 adjacent output pixels x 4 channels in registers
// and slides the 7-tap row over 13 LDS reads (49 FMA x 4 channels per 13+7 ds_read_b128).
// The input halo tile and the 49x32 weight slice are staged through LDS once per workgroup.
#include "acx_internal.h"

namespace acx {

constexpr int kDwSlice = 32;      // channels per workgroup (8 lanes x float4)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Streaming form: a workgroup owns (clip, 32-channel slice, column strip of TW pixels, segment of row tiles)
// and walks DOWN the image.  An LDS ring of TH+6 input rows is kept; each step computes TH output rows from
// the ring while the next TH input rows are already in flight to registers (issued before the FMAs, written
// into the ring slots of the TH oldest rows after them), so every input row is fetched once per strip
// (the first tile-only version re-read its 6 halo rows per tile: FETCH_SIZE 1.93x the algorithmic bytes,
// profiles/r01_c_pmc_per_kernel.csv) and HBM latency hides under the arithmetic.
template <int TW, int TH>
struct DwCfg {
    static constexpr int WT = 7;
    static constexpr int kStrips = TW / WT;
    static constexpr int kThreads = kStrips * 8 * TH;
    static constexpr int kCols = TW + 6;
    static constexpr int kRing = TH + 6;
    static constexpr int kRowF4 = kCols * 8;                      // float4 per ring row
    static constexpr int kStepF4 = TH * kRowF4;                   // float4 fetched per step
    static constexpr int kStage = (kStepF4 + kThreads - 1) / kThreads;   // staging float4 per thread
    static constexpr size_t kLdsBytes = (size_t)(kRing * kRowF4 + 49 * 8) * 16;
};

template <int TW, int TH>
__global__ __launch_bounds__(256, 2) void dwconv7_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ wt /*[49][C]*/,
                                                      const float* __restrict__ bias, int H, int W, int C,
                                                      int tiles_w, int tiles_h, int n_seg) {
    using Cfg = DwCfg<TW, TH>;
    static_assert(Cfg::kThreads == 256, "thread mapping assumes 256 threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* ring = reinterpret_cast<f32x4*>(smem);                  // [kRing][kCols][8]
    f32x4* wl = ring + Cfg::kRing * Cfg::kRowF4;                   // [49][8]

    int bid = blockIdx.x;
    const int slice = bid % (C / kDwSlice); bid /= (C / kDwSlice);
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int seg = bid % n_seg; bid /= n_seg;
    const long long b = bid;
    const int c0 = slice * kDwSlice;
    const int w0 = tw * TW;
    const int tid = threadIdx.x;
    // balanced segments of row tiles
    const int t_begin = (int)((long long)tiles_h * seg / n_seg);
    const int t_end = (int)((long long)tiles_h * (seg + 1) / n_seg);
    if (t_begin >= t_end) return;

    for (int i = tid; i < 49 * 8; i += Cfg::kThreads)
        wl[i] = *reinterpret_cast<const f32x4*>(wt + (i >> 3) * C + c0 + 4 * (i & 7));
    const float* xb = x + b * (long long)H * W * C + c0;

    // staging coordinates of this thread's float4 #k within a step of TH rows: (row, col, quad).
    // Loads are UNCONDITIONAL on clamped (always valid) addresses and zero-padding is applied when the
    // registers are written to the ring: a per-element "load or zero" makes hipcc branch around every load
    // and wait for all of them right there, which would serialise the prefetch with the FMAs.
    int st_row[Cfg::kStage], st_off[Cfg::kStage];
    unsigned col_ok = 0;                                // bit k: column inside the image and slot in range
#pragma unroll
    for (int k = 0; k < Cfg::kStage; ++k) {
        const int i = tid + k * Cfg::kThreads;
        const int qq = i & 7;
        const int col = (i >> 3) % Cfg::kCols;
        st_row[k] = (i < Cfg::kStepF4) ? (i >> 3) / Cfg::kCols : 0;
        const int gw = w0 - 3 + col;
        const int gwc = gw < 0 ? 0 : (gw >= W ? W - 1 : gw);
        st_off[k] = gwc * C + 4 * qq;
        if (i < Cfg::kStepF4 && gw >= 0 && gw < W) col_ok |= 1u << k;
    }
    const int g_origin = t_begin * TH - 3;             // image row kept in ring row 0 of this segment
    f32x4 rg[Cfg::kStage];
#define ACX_DW_LOAD(first_row)  /* image rows first_row .. +TH-1 -> registers (row index clamped) */      \
    _Pragma("unroll") for (int k = 0; k < Cfg::kStage; ++k) {                                             \
        int gh = (first_row) + st_row[k];                                                                 \
        gh = gh < 0 ? 0 : (gh >= H ? H - 1 : gh);                                                         \
        rg[k] = *reinterpret_cast<const f32x4*>(xb + (long long)gh * W * C + st_off[k]);                  \
    }
#define ACX_DW_STORE(first_row, max_rows) /* registers -> ring rows, zero outside the image */            \
    _Pragma("unroll") for (int k = 0; k < Cfg::kStage; ++k) {                                             \
        const int i = tid + k * Cfg::kThreads;                                                            \
        if (i < Cfg::kStepF4 && st_row[k] < (max_rows)) {                                                 \
            const int gh = (first_row) + st_row[k];                                                       \
            const bool ok = ((col_ok >> k) & 1u) && gh >= 0 && gh < H;                                    \
            const int slot = (gh - g_origin) % Cfg::kRing;                                                \
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                                         \
            ring[slot * Cfg::kRowF4 + (i - st_row[k] * Cfg::kRowF4)] = ok ? rg[k] : z;                    \
        }                                                                                                 \
    }
    // prologue: image rows [g_origin, g_origin + TH + 6) in two rounds (of the second only 6 rows are kept)
    ACX_DW_LOAD(g_origin)
    ACX_DW_STORE(g_origin, TH)
    ACX_DW_LOAD(g_origin + TH)
    ACX_DW_STORE(g_origin + TH, 6)
    __syncthreads();

    const int q = tid & 7;
    const int strip = (tid >> 3) % Cfg::kStrips;
    const int r = (tid >> 3) / Cfg::kStrips;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c0 + 4 * q);

    for (int t = t_begin; t < t_end; ++t) {
        const int h0 = t * TH;
        const bool more = t + 1 < t_end;
        // next step's rows: image rows h0 + TH + 3 .. h0 + 2TH + 2  (ring rows of the TH oldest)
        const int next_first = h0 + TH + 3;
        ACX_DW_LOAD(next_first)           // (on the last tile of the segment: a discarded, in-range load)
        __builtin_amdgcn_sched_barrier(0);

        f32x4 acc[Cfg::WT];
#pragma unroll
        for (int i = 0; i < Cfg::WT; ++i) acc[i] = bv;
        int qw = q;                       // opaque per tile: keeps hipcc from hoisting all 49 weight float4
        asm volatile("" : "+v"(qw));      // (196 VGPRs) out of the tile loop -- they are re-read from LDS instead
        const int base = (h0 - 3 - g_origin + r) % Cfg::kRing;    // ring slot of input row h0 - 3 + r
        // The 7 kernel rows are software-pipelined by hand in HALF rows: the LDS reads of the next unit
        // (7 or 6 input float4, and once per row the 7 weight float4 of the next kernel row) are issued into
        // a second register set BEFORE the 49 packed FMAs of the current unit.  With two waves per SIMD there
        // is not enough TLP to cover an LDS round trip per kernel row otherwise (the rolled loop spent 55 % of
        // its wave time in s_waitcnt).  Half rows keep the double buffers at 112 VGPRs.
        f32x4 iA[7], iB[6], wA[7], wB[7];
#define ACX_DW_ROWP(ky_, p_)                                                                              \
        const f32x4* p_;                                                                                  \
        {                                                                                                 \
            int slot = base + (ky_);                                                                      \
            if (slot >= Cfg::kRing) slot -= Cfg::kRing;                                                   \
            p_ = ring + (slot * Cfg::kCols + strip * Cfg::WT) * 8 + q;                                    \
        }
#define ACX_DW_READ_W(w_, ky_) _Pragma("unroll") for (int kx = 0; kx < 7; ++kx) w_[kx] = wl[((ky_) * 7 + kx) * 8 + qw];
#define ACX_DW_READ_I0(ky_) { ACX_DW_ROWP(ky_, p0_) _Pragma("unroll") for (int j = 0; j < 7; ++j) iA[j] = p0_[j * 8]; }
#define ACX_DW_READ_I1(ky_) { ACX_DW_ROWP(ky_, p1_) _Pragma("unroll") for (int j = 0; j < 6; ++j) iB[j] = p1_[(7 + j) * 8]; }
#define ACX_DW_PIN _Pragma("unroll") for (int i = 0; i < Cfg::WT; ++i) asm volatile("" : "+v"(acc[i]));
        // opaque re-definition of acc pins the FMAs in place (plain arithmetic is otherwise sunk below later reads)
#define ACX_DW_FMA0(w_)                                                                                   \
        {                                                                                                 \
            _Pragma("unroll") for (int j = 0; j < 7; ++j)                                                 \
            _Pragma("unroll") for (int kx = 0; kx <= j; ++kx) acc[j - kx] += iA[j] * w_[kx];              \
            ACX_DW_PIN                                                                                    \
        }
#define ACX_DW_FMA1(w_)                                                                                   \
        {                                                                                                 \
            _Pragma("unroll") for (int j = 7; j < 13; ++j)                                                \
            _Pragma("unroll") for (int kx = j - 6; kx < 7; ++kx) acc[j - kx] += iB[j - 7] * w_[kx];       \
            ACX_DW_PIN                                                                                    \
        }
#define ACX_DW_TOUCH(arr_, n_) _Pragma("unroll") for (int z = 0; z < (n_); ++z) asm volatile("" :: "v"(arr_[z]));
#define ACX_DW_KROW(ky_, wc_, wn_)   /* kernel row ky_ with weights wc_; prefetches row ky_+1 into wn_ */   \
        {                                                                                                 \
            ACX_DW_READ_I1(ky_)                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            ACX_DW_FMA0(wc_)                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            ACX_DW_TOUCH(iB, 6)                                                                           \
            if ((ky_) + 1 < 7) { ACX_DW_READ_W(wn_, (ky_) + 1) ACX_DW_READ_I0((ky_) + 1) }                \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            ACX_DW_FMA1(wc_)                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if ((ky_) + 1 < 7) { ACX_DW_TOUCH(wn_, 7) ACX_DW_TOUCH(iA, 7) }                               \
        }
        ACX_DW_READ_W(wA, 0)
        ACX_DW_READ_I0(0)
        ACX_DW_KROW(0, wA, wB) ACX_DW_KROW(1, wB, wA) ACX_DW_KROW(2, wA, wB) ACX_DW_KROW(3, wB, wA)
        ACX_DW_KROW(4, wA, wB) ACX_DW_KROW(5, wB, wA) ACX_DW_KROW(6, wA, wB)
#undef ACX_DW_ROWP
#undef ACX_DW_READ_W
#undef ACX_DW_READ_I0
#undef ACX_DW_READ_I1
#undef ACX_DW_PIN
#undef ACX_DW_FMA0
#undef ACX_DW_FMA1
#undef ACX_DW_TOUCH
#undef ACX_DW_KROW
        const int h = h0 + r;
        if (h < H) {
            float* yp = y + ((b * H + h) * (long long)W + w0 + strip * Cfg::WT) * C + c0 + 4 * q;
#pragma unroll
            for (int i = 0; i < Cfg::WT; ++i) *reinterpret_cast<f32x4*>(yp + (long long)i * C) = acc[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            __syncthreads();                                       // everyone is done reading the ring
            asm volatile("" : "+v"(rg[0]));                        // keep the vmcnt wait down here
            ACX_DW_STORE(next_first, TH)
            __syncthreads();
        }
    }
#undef ACX_DW_LOAD
#undef ACX_DW_STORE
}

// Per-row LayerNorm statistics over C channels (biased variance, eps inside the sqrt --
// convnext.py:537-540 / F.layer_norm): stats[row] = (mean, rstd).  G = C/12 lanes per row, each
// lane holds 3 float4; reductions are xor-shuffles inside the G-lane group (G = 8..64).
// NORMALIZE: write (x - mean) * rstd instead (input of the downsample convs, convnext.py:230-235).
template <int G, bool NORMALIZE>
__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                       long long rows, float eps) {
    constexpr int C = G * 12;
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x % G;
    const int sub = threadIdx.x / G;
    const long long stride = (long long)gridDim.x * RPB;
    const long long iters = (rows + stride - 1) / stride;
    for (long long it = 0; it < iters; ++it) {
        long long row = it * stride + (long long)blockIdx.x * RPB + sub;
        const bool valid = row < rows;
        if (!valid) row = rows - 1;
        const float4* p = reinterpret_cast<const float4*>(x + row * C);
        float4 v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = p[g + G * k];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = v[k].x - mean, b2 = v[k].y - mean, c2 = v[k].z - mean, e = v[k].w - mean;
            d += (a * a + b2 * b2) + (c2 * c2 + e * e);
        }
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) d += __shfl_xor(d, o);
        const float rstd = 1.0f / sqrtf(d * (1.0f / C) + eps);
        if (NORMALIZE) {      // `stats` is the (rows, C) output: (x - mean) * rstd, affine folded downstream
            if (valid) {
                float4* o = reinterpret_cast<float4*>(stats + row * C);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    o[g + G * k] = make_float4((v[k].x - mean) * rstd, (v[k].y - mean) * rstd,
                                               (v[k].z - mean) * rstd, (v[k].w - mean) * rstd);
            }
        } else if (valid && g == 0) {
            *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(mean, rstd);
        }
    }
}

template <bool NORMALIZE>
static int launch_rows(acx_ctx* c, const float* x, float* out, int64_t M, int C, hipStream_t s) {
    const int G = C / 12;
    long long blocks = (M + (256 / G) - 1) / (256 / G);
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(c, ACX_K_ROWSTATS, s);
    dim3 grid((unsigned)blocks), blk(256);
    switch (C) {
        case 96: rowstats_kernel<8, NORMALIZE><<<grid, blk, 0, s>>>(x, out, M, 1e-6f); break;
        case 192: rowstats_kernel<16, NORMALIZE><<<grid, blk, 0, s>>>(x, out, M, 1e-6f); break;
        case 384: rowstats_kernel<32, NORMALIZE><<<grid, blk, 0, s>>>(x, out, M, 1e-6f); break;
        case 768: rowstats_kernel<64, NORMALIZE><<<grid, blk, 0, s>>>(x, out, M, 1e-6f); break;
        default: ACX_FAIL(ACX_ERR_SHAPE, "rowstats: unsupported channel count %d", C);
    }
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

int launch_rowstats(acx_ctx* c, const float* x, float* stats, int64_t M, int C, hipStream_t s) {
    return launch_rows<false>(c, x, stats, M, C, s);
}

int launch_layernorm_rows(acx_ctx* c, const float* x, float* out, int64_t M, int C, hipStream_t s) {
    return launch_rows<true>(c, x, out, M, C, s);
}

template <int TW, int TH>
static int launch_dw_cfg(const BlockW& w, int C, const float* x, float* y, int B, int H, int W, hipStream_t s) {
    using Cfg = DwCfg<TW, TH>;
    static bool attr_set = false;
    if (!attr_set) {
        ACX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv7_kernel<TW, TH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::kLdsBytes));
        attr_set = true;
    }
    const int tiles_w = W / TW, tiles_h = (H + TH - 1) / TH;
    const long long columns = (long long)B * tiles_w * (C / kDwSlice);
    // enough workgroups for ~4 rounds on 256 CUs x 2 resident, but segments of at least 2 row tiles
    int n_seg = (int)((2048 + columns - 1) / columns);
    if (n_seg > tiles_h / 2) n_seg = tiles_h / 2;
    if (n_seg < 1) n_seg = 1;
    const long long blocks = columns * n_seg;
    dwconv7_kernel<TW, TH><<<dim3((unsigned)blocks), dim3(Cfg::kThreads), Cfg::kLdsBytes, s>>>(
        x, y, w.dw, w.dwb, H, W, C, tiles_w, tiles_h, n_seg);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

int launch_dwconv(acx_ctx* c, const BlockW& w, int C, const float* x, float* y, float* stats, int B, int H,
                  int W, hipStream_t s) {
    {
        ProfScope ps(c, ACX_K_DWCONV, s);
        int rc;
        switch (W) {
            case 56: rc = launch_dw_cfg<28, 8>(w, C, x, y, B, H, W, s); break;
            case 28: rc = launch_dw_cfg<28, 8>(w, C, x, y, B, H, W, s); break;
            case 14: rc = launch_dw_cfg<14, 16>(w, C, x, y, B, H, W, s); break;
            case 7: rc = launch_dw_cfg<7, 32>(w, C, x, y, B, H, W, s); break;   // 31x7 image: one step
            default: ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: unsupported width %d (expected 56/28/14/7)", W);
        }
        ACX_TRY(rc);
    }
    if (stats) ACX_TRY(launch_rowstats(c, y, stats, (int64_t)B * H * W, C, s));
    return ACX_OK;
}

}  // namespace acx
