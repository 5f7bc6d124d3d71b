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

template <int TW, int TH>
struct DwCfg {
    static constexpr int WT = 7;
    static constexpr int kStrips = TW / WT;
    static constexpr int kThreads = kStrips * 8 * TH;
    static constexpr int kCols = TW + 6;
    static constexpr int kRows = TH + 6;
    static constexpr int kTileF4 = kRows * kCols * 8;
    static constexpr size_t kLdsBytes = (size_t)(kTileF4 + 49 * 8) * 16;
};

template <int TW, int TH>
__global__ __launch_bounds__(256) void dwconv7_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ wt /*[49][C]*/,
                                                      const float* __restrict__ bias, int H, int W, int C,
                                                      int tiles_w, int tiles_h) {
    using Cfg = DwCfg<TW, TH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* tile = reinterpret_cast<float4*>(smem);                 // [kRows][kCols][8]
    float4* wl = tile + Cfg::kTileF4;                               // [49][8]

    int bid = blockIdx.x;
    const int slice = bid % (C / kDwSlice); bid /= (C / kDwSlice);
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const long long b = bid;
    const int c0 = slice * kDwSlice;
    const int h0 = th * TH, w0 = tw * TW;
    const int tid = threadIdx.x;

    for (int i = tid; i < 49 * 8; i += Cfg::kThreads)
        wl[i] = *reinterpret_cast<const float4*>(wt + (i >> 3) * C + c0 + 4 * (i & 7));
    const float* xb = x + b * (long long)H * W * C;
    for (int i = tid; i < Cfg::kTileF4; i += Cfg::kThreads) {
        const int q = i & 7;
        const int col = (i >> 3) % Cfg::kCols;
        const int row = (i >> 3) / Cfg::kCols;
        const int gh = h0 - 3 + row, gw = w0 - 3 + col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gh >= 0 && gh < H && gw >= 0 && gw < W)
            v = *reinterpret_cast<const float4*>(xb + ((long long)gh * W + gw) * C + c0 + 4 * q);
        tile[i] = v;
    }
    __syncthreads();

    const int q = tid & 7;
    const int strip = (tid >> 3) % Cfg::kStrips;
    const int r = (tid >> 3) / Cfg::kStrips;
    const float4 bv = *reinterpret_cast<const float4*>(bias + c0 + 4 * q);
    float4 acc[Cfg::WT];
#pragma unroll
    for (int i = 0; i < Cfg::WT; ++i) acc[i] = bv;

#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        float4 wk[7];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) wk[kx] = wl[(ky * 7 + kx) * 8 + q];
        const float4* rowp = tile + ((r + ky) * Cfg::kCols + strip * Cfg::WT) * 8 + q;
#pragma unroll
        for (int j = 0; j < Cfg::WT + 6; ++j) {
            const float4 in = rowp[j * 8];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int i = j - kx;
                if (i >= 0 && i < Cfg::WT) {
                    acc[i].x = fmaf(in.x, wk[kx].x, acc[i].x);
                    acc[i].y = fmaf(in.y, wk[kx].y, acc[i].y);
                    acc[i].z = fmaf(in.z, wk[kx].z, acc[i].z);
                    acc[i].w = fmaf(in.w, wk[kx].w, acc[i].w);
                }
            }
        }
    }
    const int h = h0 + r;
    if (h < H) {
        float* yp = y + ((b * H + h) * (long long)W + w0 + strip * Cfg::WT) * C + c0 + 4 * q;
#pragma unroll
        for (int i = 0; i < Cfg::WT; ++i) *reinterpret_cast<float4*>(yp + (long long)i * C) = acc[i];
    }
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
    const long long blocks = (long long)B * tiles_h * tiles_w * (C / kDwSlice);
    dwconv7_kernel<TW, TH><<<dim3((unsigned)blocks), dim3(Cfg::kThreads), Cfg::kLdsBytes, s>>>(
        x, y, w.dw, w.dwb, H, W, C, tiles_w, tiles_h);
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
            case 7: rc = launch_dw_cfg<7, 32>(w, C, x, y, B, H, W, s); break;
            default: ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: unsupported width %d (expected 56/28/14/7)", W);
        }
        ACX_TRY(rc);
    }
    if (stats) ACX_TRY(launch_rowstats(c, y, stats, (int64_t)B * H * W, C, s));
    return ACX_OK;
}

}  // namespace acx
