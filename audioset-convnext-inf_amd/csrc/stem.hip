// K2 -- stem: Conv2d(1,96,k=(4,4),s=(4,4),padding=(4,0)) + LayerNorm(96) channels_first
// (convnext.py:688-691 installed at :707; LayerNorm :227, math :536-541).
//
// in : bn0'd log-mel (B,T,224) fp32.  The stem's zero padding is applied AFTER bn0
//      (convnext.py:304-306 then :690), so padded time rows are literal zeros here.
// out: NHWC (B,H0,56,96), H0 = (T+8-4)/4+1.  Output row h reads time rows 4h-4 .. 4h-1.
// 16->96 is too thin for MFMA: VALU patch-embed.  A workgroup takes whole output rows (b, h): the 4 x 224 input
// floats of the row are fetched ONCE, coalesced, into LDS (the first version let every lane of a pixel load the
// same 4 float4 itself: 64-lane broadcast loads keep the texture-address unit as busy as distinct ones, and it,
// not HBM, set the pace -- 2.4 TB/s); each 32-lane group then walks 7 of the row's 56 pixels, 24 of its lanes
// owning 4 consecutive channels each, so a pixel leaves as ONE 384-byte run of float4 stores.  LayerNorm
// statistics by xor-shuffles inside the 32-lane group (idle lanes contribute zeros).  HBM-bound, and 86 % of
// the bytes are writes: 0.90 MB in + 5.42 MB out per 10 s clip; 2.9 TB/s at B = 64 (keeping all seven pixels of a
// group in registers to overlap their shuffle chains changes nothing: it is not latency).
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

constexpr int kStemGroups = 8;                 // 32-lane groups per workgroup; 56 pixels / 8 = 7 pixels per group and row
constexpr int kStemRowF4 = kMels / 4;          // 56 float4 per input row = one per output pixel

// OBF: the output tensor is bf16 (ACX_PREC_BF16_ACT): a lane's four channels leave as one 8-byte store
template <bool OBF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void stem_kernel(const float* __restrict__ in, int T, int H0, long long nrows,
                                                   const float* __restrict__ w /*[96][16]*/,
                                                   const float* __restrict__ bias, const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, void* __restrict__ out_) {
    __shared__ float4 patch[2][4][kStemRowF4];          // double-buffered: one barrier per row
    const int tid = threadIdx.x;
    const int l32 = tid & 31;
    const int grp = tid >> 5;
    const bool owner = l32 < 24;
    const int ch = owner ? 4 * l32 : 92;                // idle lanes compute on valid addresses and are masked out
    float wr[4][16], br[4], gw[4], gb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(w + (ch + c) * 16 + 4 * k4);
            wr[c][4 * k4 + 0] = v.x; wr[c][4 * k4 + 1] = v.y; wr[c][4 * k4 + 2] = v.z; wr[c][4 * k4 + 3] = v.w;
        }
        br[c] = bias[ch + c];
        gw[c] = lnw[ch + c];
        gb[c] = lnb[ch + c];
    }
    const float keep = owner ? 1.0f : 0.0f;
    // staging: thread i < 224 fetches float4 (ky = i / 56, w = i % 56) of the current row
    const int sky = tid / kStemRowF4, sw = tid - sky * kStemRowF4;
    const bool stager = tid < 4 * kStemRowF4;
    auto fetch = [&](long long row) -> float4 {
        const int h = (int)(row % H0);
        const long long b = row / H0;
        int t = 4 * h - 4 + sky;
        const bool rok = stager && t >= 0 && t < T;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);           // clamped address, zeroed below (no branchy loads)
        float4 v = *reinterpret_cast<const float4*>(in + (b * T + t) * kMels + 4 * (stager ? sw : 0));
        if (!rok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    };
    long long row = blockIdx.x;
    if (row >= nrows) return;
    // the rows of the next TWO iterations are in flight while this one is computed: a row is ~1.5 us of arithmetic per workgroup,
    // an HBM round trip under load is longer (with one row of look-ahead the launch waited for its input: wait_any 0.57).  The
    // loop is unrolled by two so that each of the two staging registers is written by a load and read two iterations later.
    const long long g = gridDim.x;
    float4 pa = fetch(row);
    float4 pb = row + g < nrows ? fetch(row + g) : make_float4(0.f, 0.f, 0.f, 0.f);
    int buf = 0;
    auto one_row = [&](const long long row, float4& stage) __attribute__((always_inline)) {
        if (stager) patch[buf][sky][sw] = stage;
        // a bare barrier behind the LDS writes: __syncthreads() also waits vmcnt(0) -- for the previous row's 347 MB-per-launch
        // output stores and for the row just requested
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // this row's patches visible; the other buffer is free again
        if (row + 2 * g < nrows) stage = fetch(row + 2 * g);
        float* orow = reinterpret_cast<float*>(out_) + row * (long long)(kStemW * 96) + ch;
        __bf16* orow_b = reinterpret_cast<__bf16*>(out_) + row * (long long)(kStemW * 96) + ch;
        // not unrolled, and the kernel held to 128 registers: four waves per SIMD instead of three (unrolled, hipcc hoists the LDS
        // reads of all seven pixels: 132-146 registers) -- 149 -> 130 us at B = 64
#pragma unroll 1
        for (int i = 0; i < kStemW / kStemGroups; ++i) {
            const int px = grp + kStemGroups * i;
            float acc[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const float4 p = patch[buf][ky][px];    // same address for the whole group: LDS broadcast
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = fmaf(p.x, wr[c][4 * ky + 0], acc[c]);
                    acc[c] = fmaf(p.y, wr[c][4 * ky + 1], acc[c]);
                    acc[c] = fmaf(p.z, wr[c][4 * ky + 2], acc[c]);
                    acc[c] = fmaf(p.w, wr[c][4 * ky + 3], acc[c]);
                }
            }
            const float mean = half_sum32(((acc[0] + acc[1]) + (acc[2] + acc[3])) * keep) * (1.0f / 96.0f);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] -= mean;
            const float var = half_sum32(((acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3])) * keep) * (1.0f / 96.0f);
            const float rstd = 1.0f / sqrtf(var + 1e-6f);
            if (owner) {
                const float o0 = fmaf(acc[0] * rstd, gw[0], gb[0]), o1 = fmaf(acc[1] * rstd, gw[1], gb[1]),
                            o2 = fmaf(acc[2] * rstd, gw[2], gb[2]), o3 = fmaf(acc[3] * rstd, gw[3], gb[3]);
                if (OBF) *reinterpret_cast<uint2*>(orow_b + px * 96) = uint2{acx_pack_bf16x2(o0, o1), acx_pack_bf16x2(o2, o3)};
                else *reinterpret_cast<float4*>(orow + px * 96) = make_float4(o0, o1, o2, o3);
            }
        }
        buf ^= 1;
    };
    for (; row < nrows; row += 2 * g) {
        one_row(row, pa);
        if (row + g < nrows) one_row(row + g, pb);
    }
}

int launch_stem(acx_ctx* c, const float* in, int B, int T, int H0, void* out, hipStream_t s, bool act_bf16) {
    const long long nrows = (long long)B * H0;
    long long blocks = nrows < 2048 ? nrows : 2048;     // 8 resident workgroups per CU, every workgroup walks ~8 rows at B = 64
    ProfScope ps(c, ACX_K_STEM, s);
    if (act_bf16)
        launch_kernel(&stem_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, in, T, H0, nrows, c->d_stem_w, c->d_stem_b,
                                                                        c->d_stem_lnw, c->d_stem_lnb, out);
    else
        launch_kernel(&stem_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, in, T, H0, nrows, c->d_stem_w, c->d_stem_b,
                                                                         c->d_stem_lnw, c->d_stem_lnb, out);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

}  // namespace acx
