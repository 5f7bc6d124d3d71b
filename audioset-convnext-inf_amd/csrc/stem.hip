// K2 -- stem: Conv2d(1,96,k=(4,4),s=(4,4),padding=(4,0)) + LayerNorm(96) channels_first
// (convnext.py:688-691 installed at :707; LayerNorm :227, math :536-541).
//
// in : bn0'd log-mel (B,T,224) fp32.  The stem's zero padding is applied AFTER bn0
//      (convnext.py:304-306 then :690), so padded time rows are literal zeros here.
// out: NHWC (B,H0,56,96), H0 = (T+8-4)/4+1.  Output row h reads time rows 4h-4 .. 4h-1.
// 16->96 is too thin for MFMA: VALU patch-embed, 32 lanes x 3 channels per pixel (2 pixels per
// wave64), LayerNorm statistics by xor-shuffles inside each 32-lane half.  HBM-bound:
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

__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ in, int T, int H0, long long npix,
                                                   const float* __restrict__ w /*[96][16]*/,
                                                   const float* __restrict__ bias, const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, float* __restrict__ out) {
    const int l32 = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;             // 8 pixels per block iteration
    const int c0 = 3 * l32;
    float wr[3][16], br[3], gw[3], gb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int k = 0; k < 16; ++k) wr[c][k] = w[(c0 + c) * 16 + k];
        br[c] = bias[c0 + c];
        gw[c] = lnw[c0 + c];
        gb[c] = lnb[c0 + c];
    }
    const long long stride = (long long)gridDim.x * 8;
    const long long iters = (npix + stride - 1) / stride;      // uniform trip count (shuffles need full waves)
    for (long long it = 0; it < iters; ++it) {
        long long p = it * stride + (long long)blockIdx.x * 8 + sub;
        const bool valid = p < npix;
        if (!valid) p = npix - 1;
        const int wcol = (int)(p % kStemW);
        const long long bh = p / kStemW;
        const int h = (int)(bh % H0);
        const long long b = bh / H0;
        float acc[3] = {br[0], br[1], br[2]};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int t = 4 * h - 4 + ky;
            if (t >= 0 && t < T) {
                const float4 x = *reinterpret_cast<const float4*>(in + (b * T + t) * kMels + 4 * wcol);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc[c] = fmaf(x.x, wr[c][4 * ky + 0], acc[c]);
                    acc[c] = fmaf(x.y, wr[c][4 * ky + 1], acc[c]);
                    acc[c] = fmaf(x.z, wr[c][4 * ky + 2], acc[c]);
                    acc[c] = fmaf(x.w, wr[c][4 * ky + 3], acc[c]);
                }
            }
        }
        const float mean = half_sum32(acc[0] + acc[1] + acc[2]) * (1.0f / 96.0f);
        const float d0 = acc[0] - mean, d1 = acc[1] - mean, d2 = acc[2] - mean;
        const float var = half_sum32(d0 * d0 + d1 * d1 + d2 * d2) * (1.0f / 96.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-6f);
        if (valid) {
            float* o = out + p * 96 + c0;
            o[0] = fmaf(d0 * rstd, gw[0], gb[0]);
            o[1] = fmaf(d1 * rstd, gw[1], gb[1]);
            o[2] = fmaf(d2 * rstd, gw[2], gb[2]);
        }
    }
}

int launch_stem(acx_ctx* c, const float* in, int B, int T, int H0, float* out, hipStream_t s) {
    const long long npix = (long long)B * H0 * kStemW;
    long long blocks = (npix + 7) / 8;
    if (blocks > 8192) blocks = 8192;
    ProfScope ps(c, ACX_K_STEM, s);
    stem_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(in, T, H0, npix, c->d_stem_w, c->d_stem_b,
                                                              c->d_stem_lnw, c->d_stem_lnb, out);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

}  // namespace acx
