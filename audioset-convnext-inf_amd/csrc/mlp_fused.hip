// K4f -- one launch for LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual
// (Block.forward after the depthwise conv, convnext.py:77-86) for the wide-and-shallow stages
// (C = 96, 192), where the unfused pair of GEMMs is dominated by writing and re-reading the (M, 4C)
// hidden activation (1.39 GB per block at batch 64 in stage 0) and by the GELU/store epilogue.
// The hidden activation never leaves the register file:
//
//   each wave owns 32 pixels for the whole block.  Everything is computed TRANSPOSED so that the result
//   tile of the first product is directly the B operand of the second (v_mfma_f32_32x32x2_f32 D layout:
//   column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)):
//     phase 1   X^T[32 hidden x 32 px] = W1c[32 x C] . LN(y)^T[C x 32 px]      C/2 MFMAs, B operand = the
//               wave's normalised activations, resident in C/2 VGPRs for the whole kernel
//     GELU      on the 16 accumulator registers (+ bias, pre-loaded as the initial accumulator)
//     phase 2   out^T[C x 32 px] += W2c[C x 32 hidden] . X^T                    C/2 MFMAs; register r of X^T is
//               used as-is for k-step r, which contracts hidden rows {R(r), R(r)+4}; the A operand (W2c)
//               is read with the matching k order: 4 consecutive hidden values per ds_read_b128.
//   The 4C hidden units are processed in chunks of 32; per chunk the workgroup's waves share one image of
//   [W1c | W2c] (256*C bytes) in LDS, double-buffered and filled by LDS-DMA from a chunk-major repack of
//   the weights made at acx_finalize (XOR swizzle on the source address, see gemm.hip).
// LayerNorm statistics are computed in-kernel from the wave's own rows (two-pass, in registers).
// HBM traffic per block: read y, read x, write x (3*C*H*W*4 B) -- the algorithmic minimum for this split.
#include "acx_internal.h"

namespace acx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_erf_f(float v) {     // see gemm.hip: A&S 7.1.26 erf, 14 VALU
    const float av = fabsf(v);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678f, av, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(v * v * -0.72134752f);
    const float q = pl * t * e;
    return fmaf(-0.5f * av, q, fmaxf(v, 0.0f));
}

template <int C>
struct FusedCfg {
    static constexpr int kWaves = 4;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kPix = kWaves * 32;                 // pixels per workgroup
    static constexpr int kChunks = 4 * C / 32;               // hidden chunks of 32
    static constexpr int kHalfBytes = 128 * C;               // one [32][C] (or [C][32]) fp32 image
    static constexpr int kPieces = kHalfBytes / 1024 / kWaves;    // 1-KB DMA pieces per wave per image (3 / 6)
    static constexpr int kRowChunks = C / 4;                 // 16-B chunks per W1c row
    static constexpr size_t kLdsBytes = 4 * (size_t)kHalfBytes + 4 * C * 4;   // W1 ring[2] + W2 ring[2] + bias1
    __device__ static int swz1(int row) { return (C == 96) ? ((row >> 1) & 7) : (row & 15); }
};

// Software pipeline over hidden chunks j (32 hidden units each), per wave:
//   stage A:  X(j+1) = bias + W1c(j+1) . act^T     C/2 chained MFMAs      } independent streams, issued
//             G(j)   = gelu(X(j))                   16 x 14 VALU           } interleaved (VALU under MFMA)
//   stage B:  out   += W2c(j) . G(j)                C/2 MFMAs
// W1c and W2c live in two 2-deep LDS rings filled by LDS-DMA: iteration j issues W1c(j+2) and W2c(j+1).
template <int C>
__global__ __launch_bounds__(FusedCfg<C>::kThreads) void mlp_fused_kernel(
    const float* __restrict__ y, float* __restrict__ x, const float* __restrict__ wpack /*[chunks][64*C]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M) {
    using Cfg = FusedCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1buf = smem;                                                  // [2][kHalfBytes]  rows = hidden
    char* w2buf = smem + 2 * Cfg::kHalfBytes;                            // [2][kHalfBytes]  rows = out channel
    float* b1s = reinterpret_cast<float*>(smem + 4 * Cfg::kHalfBytes);   // [4C]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const long long pix0 = (long long)blockIdx.x * Cfg::kPix + wave * 32;
    long long mrow = pix0 + l31;
    const bool valid = mrow < M;
    if (!valid) mrow = M - 1;

    // ---- LDS-DMA source offsets (floats, relative to the chunk block [W1c | W2c]) for this lane ----------
    int src1[Cfg::kPieces], src2[Cfg::kPieces];
#pragma unroll
    for (int k = 0; k < Cfg::kPieces; ++k) {
        const int idx = (wave * Cfg::kPieces + k) * 64 + lane;           // linear 16-B slot in the LDS image
        const int r1 = idx / Cfg::kRowChunks, p1 = idx - r1 * Cfg::kRowChunks;
        src1[k] = (r1 * Cfg::kRowChunks + (p1 ^ Cfg::swz1(r1))) * 4;
        const int r2 = idx >> 3, p2 = idx & 7;
        src2[k] = 32 * C + (r2 * 8 + (p2 ^ ((r2 >> 1) & 7))) * 4;
    }
#define ACX_DMA(srcv, j, dstbase)                                                                               \
    {                                                                                                           \
        const float* cb = wpack + (long long)(j) * (64 * C);                                                    \
        _Pragma("unroll") for (int k = 0; k < Cfg::kPieces; ++k)                                                \
            __builtin_amdgcn_global_load_lds(                                                                   \
                (const __attribute__((address_space(1))) void*)(cb + srcv[k]),                                  \
                (__attribute__((address_space(3))) void*)((dstbase) + (wave * Cfg::kPieces + k) * 1024), 16, 0, 0); \
    }
    ACX_DMA(src1, 0, w1buf);
    ACX_DMA(src2, 0, w2buf);
    ACX_DMA(src1, 1, w1buf + Cfg::kHalfBytes);
    for (int i = tid; i < 4 * C; i += Cfg::kThreads) b1s[i] = b1[i];

    // ---- this wave's activations: lane (px = l31, half hh) holds channels [hh*C/2, hh*C/2 + C/2) ------
    float act[C / 2];
    {
        const float* yp = y + mrow * C + hh * (C / 2);
#pragma unroll
        for (int i = 0; i < C / 8; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(yp + 4 * i);
            act[4 * i + 0] = v.x; act[4 * i + 1] = v.y; act[4 * i + 2] = v.z; act[4 * i + 3] = v.w;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < C / 2; ++i) s += act[i];
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < C / 2; ++i) { const float t = act[i] - mean; d = fmaf(t, t, d); }
        d += __shfl_xor(d, 32);
        const float rstd = 1.0f / sqrtf(d * (1.0f / C) + 1e-6f);
#pragma unroll
        for (int i = 0; i < C / 2; ++i) act[i] = (act[i] - mean) * rstd;
    }

    f32x16 acc[C / 32];
#pragma unroll
    for (int t = 0; t < C / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // fragment read offsets (bytes)
    const int sw1 = Cfg::swz1(l31);
    const int w1row = l31 * (4 * C);
    const int sw2 = (l31 >> 1) & 7;
    const int w2row = l31 * 128;

#define ACX_PHASE1(Xv, j, w1p)                                                                                  \
    {                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * (j) + 8 * q + 4 * hh);                  \
            Xv[4 * q + 0] = bq[0]; Xv[4 * q + 1] = bq[1]; Xv[4 * q + 2] = bq[2]; Xv[4 * q + 3] = bq[3];         \
        }                                                                                                       \
        _Pragma("unroll") for (int t = 0; t < C / 8; ++t) {                                                     \
            const f32x4 a = *reinterpret_cast<const f32x4*>((w1p) + w1row + (((hh * (C / 8) + t) ^ sw1) << 4)); \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                       \
                Xv = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], act[4 * t + e], Xv, 0, 0, 0);                   \
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
        if (j + 1 < Cfg::kChunks) {
            const char* w1p = w1buf + ((j + 1) & 1) * Cfg::kHalfBytes;
            ACX_PHASE1(Xn, j + 1, w1p)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[r] = gelu_erf_f(X[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { X[r] = gelu_erf_f(X[r]); Xn[r] = 0.f; }
        }
        __builtin_amdgcn_sched_barrier(0);
        const char* w2p = w2buf + (j & 1) * Cfg::kHalfBytes + w2row;
        // W2c fragments double-buffered in two NAMED registers with sched_barriers between the read of
        // group i+1 and the 4 MFMAs of group i: hipcc otherwise reuses one register quad and waits on every
        // ds_read right in front of its (dependent-chain) MFMAs -- an LDS round trip per 4 MFMAs.
#define ACX_W2_READ(i_) (*reinterpret_cast<const f32x4*>(w2p + ((i_) >> 2) * 4096 + (((2 * ((i_) & 3) + hh) ^ sw2) << 4)))
#define ACX_W2_MFMA(i_, a_)                                                                                    \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                           \
            acc[(i_) >> 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[e], X[4 * ((i_) & 3) + e], acc[(i_) >> 2], 0, 0, 0);
        f32x4 a0 = ACX_W2_READ(0), a1;
#pragma unroll
        for (int i = 0; i < C / 8; i += 2) {
            a1 = ACX_W2_READ(i + 1);
            __builtin_amdgcn_sched_barrier(0);
            ACX_W2_MFMA(i, a0)
            __builtin_amdgcn_sched_barrier(0);
            // With an LDS-DMA in flight hipcc only emits lgkmcnt(0): consume a1 HERE (its read is 4 MFMAs old)
            // so that the wait does not also cover the read issued next.
            asm volatile("" :: "v"(a1));
            if (i + 2 < C / 8) a0 = ACX_W2_READ(i + 2);
            __builtin_amdgcn_sched_barrier(0);
            ACX_W2_MFMA(i + 1, a1)
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < C / 8) asm volatile("" :: "v"(a0));
        }
#undef ACX_W2_READ
#undef ACX_W2_MFMA
        X = Xn;
        __syncthreads();
    }
#undef ACX_DMA
#undef ACX_PHASE1

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
                v.x += acc[t][4 * q + 0] + bb.x;
                v.y += acc[t][4 * q + 1] + bb.y;
                v.z += acc[t][4 * q + 2] + bb.z;
                v.w += acc[t][4 * q + 3] + bb.w;
                *reinterpret_cast<float4*>(xp + c) = v;
            }
        }
    }
}

template <int C>
static int launch_fused_cfg(const BlockW& w, const float* y, float* x, long long M, hipStream_t s) {
    using Cfg = FusedCfg<C>;
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_kernel<C>, Cfg::kLdsBytes));
    const long long blocks = (M + Cfg::kPix - 1) / Cfg::kPix;
    launch_kernel(&mlp_fused_kernel<C>, dim3((unsigned)blocks), dim3(Cfg::kThreads), Cfg::kLdsBytes, s, y, x, w.wpack, w.b1, w.b2, M);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

bool mlp_fused_supported(int C) { return C == 96 || C == 192; }

int launch_mlp_fused(acx_ctx* c, const BlockW& w, int C, const float* y, float* x, long long M, hipStream_t s) {
    if (!w.wpack) ACX_FAIL(ACX_ERR_STATE, "fused MLP: chunk-major weights were not packed for C=%d", C);
    ProfScope ps(c, ACX_K_MLP_FUSED, s);
    if (C == 96) return launch_fused_cfg<96>(w, y, x, M, s);
    if (C == 192) return launch_fused_cfg<192>(w, y, x, M, s);
    ACX_FAIL(ACX_ERR_SHAPE, "fused MLP: unsupported channel count %d", C);
}

}  // namespace acx
