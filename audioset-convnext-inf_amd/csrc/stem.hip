// K2 -- stem: Conv2d(1,96,k=(4,4),s=(4,4),padding=(4,0)) + LayerNorm(96) channels_first
// (convnext.py:688-691 installed at :707; LayerNorm :227, math :536-541).
//
// in : bn0'd log-mel (B,T,224) fp32.  The stem's zero padding is applied AFTER bn0
//      (convnext.py:304-306 then :690), so padded time rows are literal zeros here.
// out: NHWC (B,H0,56,96), H0 = (T+8-4)/4+1.  Output row h reads time rows 4h-4 .. 4h-1.
// 16->96 is too thin for MFMA: VALU patch-embed, 32 lanes x 3 channels per pixel (2 pixels per
// wave64; lane l owns channels l, l+32, l+64 so that every store instruction writes 128 contiguous bytes
// per pixel), LayerNorm statistics by xor-shuffles inside each 32-lane half.  HBM-bound:
// 0.90 MB in + 5.42 MB out per 10 s clip.
#include "acx_internal.h"

namespace acx {

__device__ __forceinline__ float half_sum32(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

constexpr int kStemPix = 4;     // consecutive output pixels (same row) per 32-lane group per iteration: 16
                                // independent 16-B loads in flight per lane instead of 4

__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ in, int T, int H0, long long ngroups,
                                                   const float* __restrict__ w /*[96][16]*/,
                                                   const float* __restrict__ bias, const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, float* __restrict__ out) {
    const int l32 = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;             // 8 pixel groups per block iteration
    float wr[3][16], br[3], gw[3], gb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int ch = l32 + 32 * c;
#pragma unroll
        for (int k = 0; k < 16; ++k) wr[c][k] = w[ch * 16 + k];
        br[c] = bias[ch];
        gw[c] = lnw[ch];
        gb[c] = lnb[ch];
    }
    constexpr int GPR = kStemW / kStemPix;        // groups per output row (14)
    const long long stride = (long long)gridDim.x * 8;
    const long long iters = (ngroups + stride - 1) / stride;      // uniform trip count (shuffles need full waves)
    for (long long it = 0; it < iters; ++it) {
        long long g = it * stride + (long long)blockIdx.x * 8 + sub;
        const bool valid = g < ngroups;
        if (!valid) g = ngroups - 1;
        const int wg = (int)(g % GPR);
        const long long bh = g / GPR;
        const int h = (int)(bh % H0);
        const long long b = bh / H0;
        float4 xin[4][kStemPix];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            int t = 4 * h - 4 + ky;
            const bool rok = t >= 0 && t < T;
            t = t < 0 ? 0 : (t >= T ? T - 1 : t);              // clamped address, zeroed below (no branchy loads)
            const float4* rp = reinterpret_cast<const float4*>(in + (b * T + t) * kMels + 4 * kStemPix * wg);
#pragma unroll
            for (int px = 0; px < kStemPix; ++px) {
                float4 v = rp[px];
                if (!rok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                xin[ky][px] = v;
            }
        }
        float acc[kStemPix][3];
#pragma unroll
        for (int px = 0; px < kStemPix; ++px) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a = br[c];
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    a = fmaf(xin[ky][px].x, wr[c][4 * ky + 0], a);
                    a = fmaf(xin[ky][px].y, wr[c][4 * ky + 1], a);
                    a = fmaf(xin[ky][px].z, wr[c][4 * ky + 2], a);
                    a = fmaf(xin[ky][px].w, wr[c][4 * ky + 3], a);
                }
                acc[px][c] = a;
            }
        }
        float mean[kStemPix], var[kStemPix];
#pragma unroll
        for (int px = 0; px < kStemPix; ++px) mean[px] = half_sum32(acc[px][0] + acc[px][1] + acc[px][2]) * (1.0f / 96.0f);
#pragma unroll
        for (int px = 0; px < kStemPix; ++px) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[px][c] -= mean[px];
            var[px] = half_sum32(acc[px][0] * acc[px][0] + acc[px][1] * acc[px][1] + acc[px][2] * acc[px][2]) * (1.0f / 96.0f);
        }
        if (valid) {
            float* o = out + (g * kStemPix) * 96 + l32;
#pragma unroll
            for (int px = 0; px < kStemPix; ++px) {
                const float rstd = 1.0f / sqrtf(var[px] + 1e-6f);
                o[px * 96 + 0] = fmaf(acc[px][0] * rstd, gw[0], gb[0]);
                o[px * 96 + 32] = fmaf(acc[px][1] * rstd, gw[1], gb[1]);
                o[px * 96 + 64] = fmaf(acc[px][2] * rstd, gw[2], gb[2]);
            }
        }
    }
}

int launch_stem(acx_ctx* c, const float* in, int B, int T, int H0, float* out, hipStream_t s) {
    const long long ngroups = (long long)B * H0 * (kStemW / kStemPix);
    long long blocks = (ngroups + 7) / 8;
    if (blocks > 8192) blocks = 8192;
    ProfScope ps(c, ACX_K_STEM, s);
    stem_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(in, T, H0, ngroups, c->d_stem_w, c->d_stem_b,
                                                              c->d_stem_lnw, c->d_stem_lnb, out);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

}  // namespace acx
