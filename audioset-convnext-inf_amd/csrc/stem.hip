// K2 -- stem: Conv2d(1,96,k=(4,4),s=(4,4),padding=(4,0)) + LayerNorm(96) channels_first
// (convnext.py:688-691 installed at :707; LayerNorm :227, math :536-541).
//
// in : bn0'd log-mel (B,T,224) fp32.  The stem's zero padding is applied AFTER bn0
//      (convnext.py:304-306 then :690), so padded time rows are literal zeros here.
// out: NHWC (B,H0,56,96), H0 = (T+8-4)/4+1.  Output row h reads time rows 4h-4 .. 4h-1.
// 16->96 is too thin for MFMA: VALU patch-embed.  A workgroup takes whole output rows (b, h), two at a time: the 4 x 224 input
// floats of a row are fetched ONCE, coalesced, into LDS (the first version let every lane of a pixel load the same 4 float4
// itself: 64-lane broadcast loads keep the texture-address unit as busy as distinct ones, and it, not HBM, set the pace).
// Round 4: the launch was bound by its vector instructions, not by HBM (64 v_fma_f32 + ~25 for the LayerNorm per lane and
// pixel, a quarter of the lanes idle: 84 us of pure issue under 126 us measured, and the bf16 output of `bf16a` -- half the
// bytes -- took 118).  Now a 32-lane group owns TWO adjacent pixels, every lane three channels of both: the patch image in the
// LDS interleaves the two pixels (A.x B.x A.y B.y ...), so each tap is ONE v_pk_fma_f32 on (pixel A, pixel B) with the weight
// selected by op_sel from a register pair -- 24 packed FMAs per lane and pixel instead of 64 scalar ones, no idle lanes.
// A channel still accumulates bias, then ky, kx ascending, one fused multiply-add per tap.  LayerNorm statistics by
// xor-shuffles inside the 32-lane group.  A pixel leaves as one 384-byte run (12 bytes per lane; bf16: the even lane of a pair
// stores both lanes' six channels).  0.90 MB in + 5.42 MB out per 10 s clip.
#include "acx_internal.h"

namespace acx {

typedef float stem_f32x2 __attribute__((ext_vector_type(2)));
typedef float stem_f32x4 __attribute__((ext_vector_type(4)));

// sum over the 32 lanes of a group, in every lane: four DPP steps inside the rows of 16 lanes (quad swaps, then the two mirror
// patterns: any pairing that doubles the covered set will do for a sum) and v_permlane16_swap between the two rows -- vector
// instructions only (as __shfl_xor every step was a ds_bpermute_b32: an LDS round trip, ten of them in a dependent chain per pixel)
template <int CTRL>
__device__ __forceinline__ float stem_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_sum32(float v) {
    v += stem_dpp<0xB1>(v);           // quad_perm [1,0,3,2]
    v += stem_dpp<0x4E>(v);           // quad_perm [2,3,0,1]
    v += stem_dpp<0x141>(v);          // row_half_mirror
    v += stem_dpp<0x140>(v);          // row_mirror
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // (own or the partner row's, the other)
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

constexpr int kStemGroups = 8;                 // 32-lane groups per workgroup; 2 rows x 28 pixel pairs / 8 = 7 pairs per group
constexpr int kStemRowF4 = kMels / 4;          // 56 float4 per input row = one per output pixel
constexpr int kStemPairs = kStemW / 2;         // 28

// acc (pixel A, pixel B) += patch (A, B) * w, w = the low (SEL = 0) or high (SEL = 1) half of the weight pair
template <int SEL>
__device__ __forceinline__ void stem_pkfma(stem_f32x2& acc, const stem_f32x2 p, const stem_f32x2 wpair) {
    if (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(p), "v"(wpair));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(p), "v"(wpair));
}

struct __attribute__((packed, aligned(4))) StemF3 { float a, b, c; };
struct __attribute__((packed, aligned(4))) StemU3 { unsigned a, b, c; };

// OBF: the output tensor is bf16 (ACX_PREC_BF16_ACT)
template <bool OBF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void stem_kernel(const float* __restrict__ in, int T, int H0, long long nrows,
                                                   const float* __restrict__ w /*[96][16]*/,
                                                   const float* __restrict__ bias, const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, void* __restrict__ out_) {
    // [buffer][row of the pair][ky][pixel pair][A.x B.x A.y B.y A.z B.z A.w B.w]: double-buffered, one barrier per row pair
    __shared__ __attribute__((aligned(16))) float patch[2][2][4][kStemPairs][8];
    const int tid = threadIdx.x;
    const int l32 = tid & 31;
    const int grp = tid >> 5;
    const int ch = 3 * l32;
    stem_f32x2 wp[3][8];
    float br[3], gw[3], gb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wp[c][j] = *reinterpret_cast<const stem_f32x2*>(w + (ch + c) * 16 + 2 * j);
        br[c] = bias[ch + c];
        gw[c] = lnw[ch + c];
        gb[c] = lnb[ch + c];
    }
    // staging: thread i < 224 fetches float4 (ky = i / 56, w = i % 56) of each of the pair's two rows
    const int sky = tid / kStemRowF4, sw = tid - sky * kStemRowF4;
    const bool stager = tid < 4 * kStemRowF4;
    // (row numbers fit 32 bits -- launch_stem checks: the 64-bit divisions were ~100 scalar instructions per fetch)
    auto fetch = [&](long long row64) -> float4 {
        unsigned row = (unsigned)row64;
        if (row >= (unsigned)nrows) row = (unsigned)nrows - 1;
        const unsigned bq = row / (unsigned)H0;
        const int h = (int)(row - bq * (unsigned)H0);
        const long long b = bq;
        int t = 4 * h - 4 + sky;
        const bool rok = stager && t >= 0 && t < T;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);           // clamped address, zeroed below (no branchy loads)
        float4 v = *reinterpret_cast<const float4*>(in + (b * T + t) * kMels + 4 * (stager ? sw : 0));
        if (!rok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return v;
    };
    const long long npairs = (nrows + 1) / 2;
    long long pr = blockIdx.x;
    if (pr >= npairs) return;
    // the rows of the NEXT pair are in flight while this one is computed (two rows of arithmetic ~3 us per workgroup; an HBM round
    // trip under load is shorter): two staging registers per pair, the loop unrolled by two so that each is written by a load
    // and read one iteration later
    const long long g = gridDim.x;
    float4 pa0 = fetch(2 * pr), pa1 = fetch(2 * pr + 1);
    float4 pb0 = make_float4(0.f, 0.f, 0.f, 0.f), pb1 = pb0;
    int buf = 0;
    auto one_pair = [&](const long long pr, float4& st0, float4& st1, float4& nx0, float4& nx1) __attribute__((always_inline)) {
        if (stager) {
            float* d0 = &patch[buf][0][sky][sw >> 1][sw & 1];
            float* d1 = &patch[buf][1][sky][sw >> 1][sw & 1];
            d0[0] = st0.x; d0[2] = st0.y; d0[4] = st0.z; d0[6] = st0.w;
            d1[0] = st1.x; d1[2] = st1.y; d1[4] = st1.z; d1[6] = st1.w;
        }
        // a bare barrier behind the LDS writes: __syncthreads() also waits vmcnt(0) -- for the previous pair's output stores and for
        // the rows just requested
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // this pair's patches visible; the other buffer is free again
        if (pr + g < npairs) { nx0 = fetch(2 * (pr + g)); nx1 = fetch(2 * (pr + g) + 1); }
#pragma unroll 1
        for (int i = 0; i < 2 * kStemPairs / kStemGroups; ++i) {
            const int q = grp + kStemGroups * i;                 // pixel pair 0 .. 55 of the two rows
            const int r2 = q >= kStemPairs ? 1 : 0, pp = q - r2 * kStemPairs;
            const long long row = 2 * pr + r2;
            stem_f32x2 acc[3] = {stem_f32x2{br[0], br[0]}, stem_f32x2{br[1], br[1]}, stem_f32x2{br[2], br[2]}};
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                // same address for the whole group: LDS broadcast
                const stem_f32x4 f0 = *reinterpret_cast<const stem_f32x4*>(&patch[buf][r2][ky][pp][0]);
                const stem_f32x4 f1 = *reinterpret_cast<const stem_f32x4*>(&patch[buf][r2][ky][pp][4]);
                const stem_f32x2 p0 = __builtin_shufflevector(f0, f0, 0, 1), p1 = __builtin_shufflevector(f0, f0, 2, 3);
                const stem_f32x2 p2 = __builtin_shufflevector(f1, f1, 0, 1), p3 = __builtin_shufflevector(f1, f1, 2, 3);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    stem_pkfma<0>(acc[c], p0, wp[c][2 * ky]);
                    stem_pkfma<1>(acc[c], p1, wp[c][2 * ky]);
                    stem_pkfma<0>(acc[c], p2, wp[c][2 * ky + 1]);
                    stem_pkfma<1>(acc[c], p3, wp[c][2 * ky + 1]);
                }
            }
            float o[2][3];
#pragma unroll
            for (int e = 0; e < 2; ++e) {                        // pixel A, pixel B
                float a0 = acc[0][e], a1 = acc[1][e], a2 = acc[2][e];
                const float mean = half_sum32((a0 + a1) + a2) * (1.0f / 96.0f);
                a0 -= mean; a1 -= mean; a2 -= mean;
                const float var = half_sum32((a0 * a0 + a1 * a1) + a2 * a2) * (1.0f / 96.0f);
                const float rstd = 1.0f / sqrtf(var + 1e-6f);
                o[e][0] = fmaf(a0 * rstd, gw[0], gb[0]); o[e][1] = fmaf(a1 * rstd, gw[1], gb[1]); o[e][2] = fmaf(a2 * rstd, gw[2], gb[2]);
            }
            if (row < nrows) {
                const long long pix = row * kStemW + 2 * pp;
                if (OBF) {
                    // lanes l, l ^ 1 hold channels 3 l .. + 2 and 3 l + 3 .. + 5: the even lane stores all six (12 bytes)
                    __bf16* ob = reinterpret_cast<__bf16*>(out_) + pix * 96 + ch;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned m01 = acx_pack_bf16x2(o[e][0], o[e][1]), m2 = acx_pack_bf16x2(o[e][2], 0.f);
                        const unsigned n01 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m01, 0xB1, 0xf, 0xf, true);      // lane ^ 1
                        const unsigned n2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m2, 0xB1, 0xf, 0xf, true);
                        if ((l32 & 1) == 0)
                            *reinterpret_cast<StemU3*>(ob + e * 96) = StemU3{m01, (m2 & 0xffffu) | (n01 << 16), (n01 >> 16) | (n2 << 16)};
                    }
                } else {
                    float* of = reinterpret_cast<float*>(out_) + pix * 96 + ch;
                    *reinterpret_cast<StemF3*>(of) = StemF3{o[0][0], o[0][1], o[0][2]};
                    *reinterpret_cast<StemF3*>(of + 96) = StemF3{o[1][0], o[1][1], o[1][2]};
                }
            }
        }
        buf ^= 1;
    };
    for (; pr < npairs; pr += 2 * g) {
        one_pair(pr, pa0, pa1, pb0, pb1);
        if (pr + g < npairs) one_pair(pr + g, pb0, pb1, pa0, pa1);
    }
}

int launch_stem(acx_ctx* c, const float* in, int B, int T, int H0, void* out, hipStream_t s, bool act_bf16) {
    const long long nrows = (long long)B * H0;
    if (nrows >= (1LL << 31)) ACX_FAIL(ACX_ERR_SHAPE, "stem: %lld output rows do not fit the kernel's 32-bit row arithmetic", nrows);
    const long long npairs = (nrows + 1) / 2;
    long long blocks = npairs < 2048 ? npairs : 2048;   // 8 resident workgroups per CU, every workgroup walks ~4 row pairs at B = 64
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
