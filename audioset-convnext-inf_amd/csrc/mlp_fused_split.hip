// K4fs -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86)
// in split-fp16 arithmetic (ACX_PREC_F32_SPLIT, see gemm_split.hip for the number format and its error bound).
// Same dataflow as mlp_fused.hip -- each wave owns 32 pixels, everything is computed transposed so that the
// accumulator tile of the first product is the B operand of the second, the hidden activation never leaves the
// register file, [W1c | W2c] chunk images stream through two LDS rings by LDS-DMA -- but on
// v_mfma_f32_32x32x16_f16 with every fp32 operand carried as fp16 hi + fp16 lo (three MFMAs per product):
//     phase 1   X^T[32 hidden x 32 px] = W1c[32 x C] . LN(y)^T[C x 32 px]     3 C/16 MFMAs; B operand = the wave's
//               normalised activations (x 2^11) as hi/lo halves, resident in C/2 VGPRs: lane (px, h) holds
//               channels 16s + 8h .. +7 of k-step s
//     GELU      on the 16 accumulator registers (bias pre-loaded as the initial accumulator), x 2^4, split into
//               hi/lo halves -- packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32), 13.5 VALU per element
//     phase 2   out^T[C x 32 px] += W2c[C x 32 hidden] . G                      3 C/16 MFMAs.  Lane (px, h) holds hidden
//               units 4h + 8q + e (accumulator register 4q + e); registers 8s'..8s'+7 are used as-is for k-step s',
//               i.e. MFMA k-slot (s', h, j) contracts hidden unit 16 s' + 4h + 8 (j>>2) + (j&3) -- W2c is stored
//               with the hidden units of each chunk in exactly that order (acx_finalize), so its fragments are
//               plain 16-B reads.
// HBM traffic per block: read y, read x, write x.
#include "acx_internal.h"

namespace acx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 bc2(float v) { f32x2 r; r.x = v; r.y = v; return r; }

// v = a * s;  g = gelu_erf(v) * kH (A&S 7.1.26, see gemm.hip) for two values, returned as packed fp16 hi / lo halves
struct GeluConsts { float ps, cs, ca, cb; };
__device__ __forceinline__ void gelu_split2(f32x2 a, const GeluConsts& k, unsigned& hi, unsigned& lo) {
    f32x2 av;
    av.x = __builtin_fabsf(a.x); av.y = __builtin_fabsf(a.y);
    const f32x2 den = fma2(av, bc2(k.ps), bc2(1.0f));
    f32x2 t;
    t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
    f32x2 pl = fma2(t, bc2(1.061405429f), bc2(-1.453152027f));
    pl = fma2(pl, t, bc2(1.421413741f));
    pl = fma2(pl, t, bc2(-0.284496736f));
    pl = fma2(pl, t, bc2(0.254829592f));
    const f32x2 ex = a * a * bc2(k.cs);
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(ex.x); e.y = __builtin_amdgcn_exp2f(ex.y);
    const f32x2 q = pl * t * e;
    f32x2 pos = a * bc2(k.cb);
    pos.x = __builtin_fmaxf(pos.x, 0.f); pos.y = __builtin_fmaxf(pos.y, 0.f);
    f32x2 g = fma2(av * bc2(k.ca), q, pos);
    g.x = __builtin_fminf(g.x, 65504.f); g.y = __builtin_fminf(g.y, 65504.f);
    const h2 h = __builtin_convertvector(g, h2);
    const f32x2 back = __builtin_convertvector(h, f32x2);
    const h2 l = __builtin_convertvector(g - back, h2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

template <int C>
struct FusedSCfg {
    static constexpr int kWaves = 4;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kPix = kWaves * 32;
    static constexpr int kChunks = 4 * C / 32;
    static constexpr int kHalfBytes = 128 * C;               // one [32][C] (or [C][32]) S16 image
    static constexpr int kPieces = kHalfBytes / 1024 / kWaves;
    static constexpr int kRowChunks = C / 4;                 // 16-B chunks per W1c row
    static constexpr int kSteps = C / 16;                    // k-steps of phase 1
    static constexpr int kUnits = 2 * (C / 32);              // (out tile, k-step) units of phase 2
    static constexpr size_t kLdsBytes = 4 * (size_t)kHalfBytes + 4 * C * 4;
    __device__ static int swz1(int row) { return (C == 96) ? ((row >> 1) & 7) : (row & 15); }
};

template <int C>
__global__ __launch_bounds__(FusedSCfg<C>::kThreads) void mlp_fused_split_kernel(
    const float* __restrict__ y, float* __restrict__ x, const char* __restrict__ wpack /*[chunks][256*C bytes]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, float sinv1, float sinv2) {
    using Cfg = FusedSCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1buf = smem;                                                  // [2][kHalfBytes]  rows = hidden
    char* w2buf = smem + 2 * Cfg::kHalfBytes;                            // [2][kHalfBytes]  rows = out channel
    float* b1s = reinterpret_cast<float*>(smem + 4 * Cfg::kHalfBytes);   // [4C], pre-divided by sinv1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const long long pix0 = (long long)blockIdx.x * Cfg::kPix + wave * 32;
    long long mrow = pix0 + l31;
    const bool valid = mrow < M;
    if (!valid) mrow = M - 1;

    int src1[Cfg::kPieces], src2[Cfg::kPieces];          // byte offsets inside a chunk block [W1c | W2c]
#pragma unroll
    for (int k = 0; k < Cfg::kPieces; ++k) {
        const int idx = (wave * Cfg::kPieces + k) * 64 + lane;           // linear 16-B slot in the LDS image
        const int r1 = idx / Cfg::kRowChunks, p1 = idx - r1 * Cfg::kRowChunks;
        src1[k] = (r1 * Cfg::kRowChunks + (p1 ^ Cfg::swz1(r1))) * 16;
        const int r2 = idx >> 3, p2 = idx & 7;
        src2[k] = Cfg::kHalfBytes + (r2 * 8 + (p2 ^ ((r2 >> 1) & 7))) * 16;
    }
#define ACX_DMA(srcv, j, dstbase)                                                                               \
    {                                                                                                           \
        const char* cb = wpack + (long long)(j) * (2 * Cfg::kHalfBytes);                                        \
        _Pragma("unroll") for (int k = 0; k < Cfg::kPieces; ++k)                                                \
            __builtin_amdgcn_global_load_lds(                                                                   \
                (const __attribute__((address_space(1))) void*)(cb + srcv[k]),                                  \
                (__attribute__((address_space(3))) void*)((dstbase) + (wave * Cfg::kPieces + k) * 1024), 16, 0, 0); \
    }
    ACX_DMA(src1, 0, w1buf);
    ACX_DMA(src2, 0, w2buf);
    ACX_DMA(src1, 1, w1buf + Cfg::kHalfBytes);
    {
        const float b1scale = 1.0f / sinv1;             // a power of two
        for (int i = tid; i < 4 * C; i += Cfg::kThreads) b1s[i] = b1[i] * b1scale;
    }

    // ---- this wave's activations: lane (px = l31, half hh) holds channels 16s + 8hh .. +7, s = 0..C/16-1 --------
    f32x4 acth[Cfg::kSteps], actl[Cfg::kSteps];         // 8 fp16 halves each
    {
        float a[C / 2];
        const float* yp = y + mrow * C + 8 * hh;
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(yp + 16 * s);
            const float4 v1 = *reinterpret_cast<const float4*>(yp + 16 * s + 4);
            a[8 * s + 0] = v0.x; a[8 * s + 1] = v0.y; a[8 * s + 2] = v0.z; a[8 * s + 3] = v0.w;
            a[8 * s + 4] = v1.x; a[8 * s + 5] = v1.y; a[8 * s + 6] = v1.z; a[8 * s + 7] = v1.w;
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < C / 2; ++i) sum += a[i];
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < C / 2; ++i) { const float t = a[i] - mean; d = fmaf(t, t, d); }
        d += __shfl_xor(d, 32);
        const float sc = kSplitLnScale / sqrtf(d * (1.0f / C) + 1e-6f);
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) {
            unsigned uh[4], ul[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x2 v;
                v.x = (a[8 * s + 2 * p] - mean) * sc; v.y = (a[8 * s + 2 * p + 1] - mean) * sc;
                const h2 h = __builtin_convertvector(v, h2);
                const f32x2 back = __builtin_convertvector(h, f32x2);
                const h2 l = __builtin_convertvector(v - back, h2);
                uh[p] = __builtin_bit_cast(unsigned, h);
                ul[p] = __builtin_bit_cast(unsigned, l);
            }
            acth[s] = __builtin_bit_cast(f32x4, uint4{uh[0], uh[1], uh[2], uh[3]});
            actl[s] = __builtin_bit_cast(f32x4, uint4{ul[0], ul[1], ul[2], ul[3]});
        }
    }

    f32x16 acc[C / 32];
#pragma unroll
    for (int t = 0; t < C / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int sw1 = Cfg::swz1(l31);
    const int w1row = l31 * (4 * C);
    const int sw2 = (l31 >> 1) & 7;
    const int w2row = l31 * 128;
    GeluConsts gk;
    gk.ps = 0.3275911f * 0.70710678f * sinv1;
    gk.cs = -0.72134752f * sinv1 * sinv1;
    gk.ca = -0.5f * sinv1 * kSplitHiddenScale;
    gk.cb = sinv1 * kSplitHiddenScale;

#define ACX_H8(v_) __builtin_bit_cast(h8, v_)
#define ACX_PHASE1(Xv, j, w1p)                                                                                  \
    {                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * (j) + 8 * q + 4 * hh);                  \
            Xv[4 * q + 0] = bq[0]; Xv[4 * q + 1] = bq[1]; Xv[4 * q + 2] = bq[2]; Xv[4 * q + 3] = bq[3];         \
        }                                                                                                       \
        _Pragma("unroll") for (int s = 0; s < Cfg::kSteps; ++s) {                                               \
            const f32x4 ah = *reinterpret_cast<const f32x4*>((w1p) + w1row + (((2 * (2 * s + hh)) ^ sw1) << 4));     \
            const f32x4 al = *reinterpret_cast<const f32x4*>((w1p) + w1row + (((2 * (2 * s + hh) + 1) ^ sw1) << 4)); \
            Xv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al), ACX_H8(acth[s]), Xv, 0, 0, 0);              \
            Xv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah), ACX_H8(actl[s]), Xv, 0, 0, 0);              \
            Xv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah), ACX_H8(acth[s]), Xv, 0, 0, 0);              \
        }                                                                                                       \
    }
    __syncthreads();      // W1c(0), W1c(1), W2c(0) landed (hipcc drains the LDS-DMA before the barrier); b1s visible
    f32x16 X;
    ACX_PHASE1(X, 0, w1buf)
    __syncthreads();      // every wave is done with W1 ring slot 0 before iteration 0 refills it

    for (int j = 0; j < Cfg::kChunks; ++j) {
        if (j + 2 < Cfg::kChunks) ACX_DMA(src1, j + 2, w1buf + (j & 1) * Cfg::kHalfBytes);
        if (j + 1 < Cfg::kChunks) ACX_DMA(src2, j + 1, w2buf + ((j + 1) & 1) * Cfg::kHalfBytes);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 Xn;
        f32x4 gh[2], gl[2];                 // G(j) as B-operand halves for the two k-steps of phase 2
        {
            unsigned uh[8], ul[8];
            if (j + 1 < Cfg::kChunks) {
                const char* w1p = w1buf + ((j + 1) & 1) * Cfg::kHalfBytes;
                ACX_PHASE1(Xn, j + 1, w1p)
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) Xn[r] = 0.f;
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                f32x2 a2;
                a2.x = X[2 * p]; a2.y = X[2 * p + 1];
                gelu_split2(a2, gk, uh[p], ul[p]);
            }
            gh[0] = __builtin_bit_cast(f32x4, uint4{uh[0], uh[1], uh[2], uh[3]});
            gh[1] = __builtin_bit_cast(f32x4, uint4{uh[4], uh[5], uh[6], uh[7]});
            gl[0] = __builtin_bit_cast(f32x4, uint4{ul[0], ul[1], ul[2], ul[3]});
            gl[1] = __builtin_bit_cast(f32x4, uint4{ul[4], ul[5], ul[6], ul[7]});
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* w2p = w2buf + (j & 1) * Cfg::kHalfBytes + w2row;
        // unit i = (tile t = i >> 1, k-step s' = i & 1): fragments of block b = 2s' + hh, hi chunk 2b, lo chunk 2b+1.
        // Fragments are double-buffered in two NAMED register pairs (see mlp_fused.hip).
#define ACX_W2_RD(i_, pl_) (*reinterpret_cast<const f32x4*>(w2p + ((i_) >> 1) * 4096 + (((2 * (2 * ((i_) & 1) + hh) + (pl_)) ^ sw2) << 4)))
#define ACX_W2_MFMA(i_, ah_, al_)                                                                               \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(gh[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0); \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(gl[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0); \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(gh[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0);
        f32x4 a0h = ACX_W2_RD(0, 0), a0l = ACX_W2_RD(0, 1), a1h, a1l;
#pragma unroll
        for (int i = 0; i < Cfg::kUnits; i += 2) {
            a1h = ACX_W2_RD(i + 1, 0); a1l = ACX_W2_RD(i + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            ACX_W2_MFMA(i, a0h, a0l)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" :: "v"(a1h)); asm volatile("" :: "v"(a1l));
            if (i + 2 < Cfg::kUnits) { a0h = ACX_W2_RD(i + 2, 0); a0l = ACX_W2_RD(i + 2, 1); }
            __builtin_amdgcn_sched_barrier(0);
            ACX_W2_MFMA(i + 1, a1h, a1l)
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < Cfg::kUnits) { asm volatile("" :: "v"(a0h)); asm volatile("" :: "v"(a0l)); }
        }
#undef ACX_W2_RD
#undef ACX_W2_MFMA
        X = Xn;
        __syncthreads();
    }
#undef ACX_DMA
#undef ACX_PHASE1
#undef ACX_H8

    // ---- epilogue: lane (px, hh), tile t, q: channels 32t + 8q + 4hh .. +3  ->  x = x + out + b2 ---------
    if (valid) {
        float* xp = x + mrow * C + 4 * hh;
#pragma unroll
        for (int t = 0; t < C / 32; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * t + 8 * q;
                const float4 bb = *reinterpret_cast<const float4*>(b2 + c + 4 * hh);
                float4 v = *reinterpret_cast<const float4*>(xp + c);
                v.x += fmaf(acc[t][4 * q + 0], sinv2, bb.x);
                v.y += fmaf(acc[t][4 * q + 1], sinv2, bb.y);
                v.z += fmaf(acc[t][4 * q + 2], sinv2, bb.z);
                v.w += fmaf(acc[t][4 * q + 3], sinv2, bb.w);
                *reinterpret_cast<float4*>(xp + c) = v;
            }
        }
    }
}

template <int C>
static int launch_fused_s_cfg(const BlockW& w, const float* y, float* x, long long M, hipStream_t s) {
    using Cfg = FusedSCfg<C>;
    static bool attr_set = false;
    if (!attr_set) {
        ACX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_split_kernel<C>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::kLdsBytes));
        attr_set = true;
    }
    const long long blocks = (M + Cfg::kPix - 1) / Cfg::kPix;
    mlp_fused_split_kernel<C><<<dim3((unsigned)blocks), dim3(Cfg::kThreads), Cfg::kLdsBytes, s>>>(
        y, x, reinterpret_cast<const char*>(w.wpack_s), w.b1, w.b2, M, 1.0f / (kSplitLnScale * w.w1s_scale),
        1.0f / (kSplitHiddenScale * w.w2s_scale));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

int launch_mlp_fused_split(acx_ctx* c, const BlockW& w, int C, const float* y, float* x, long long M, hipStream_t s) {
    if (!w.wpack_s) ACX_FAIL(ACX_ERR_STATE, "fused split MLP: chunk-major S16 weights were not packed for C=%d", C);
    ProfScope ps(c, ACX_K_MLP_FUSED, s);
    if (C == 96) return launch_fused_s_cfg<96>(w, y, x, M, s);
    if (C == 192) return launch_fused_s_cfg<192>(w, y, x, M, s);
    ACX_FAIL(ACX_ERR_SHAPE, "fused split MLP: unsupported channel count %d", C);
}

}  // namespace acx
