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
    static constexpr int kChunkBytes = 256 * C;              // [32][C] + [C][32] fp32
    static constexpr int kPieces = kChunkBytes / 1024 / kWaves;   // 1-KB DMA pieces per wave per chunk (6)
    static constexpr int kW1Chunks16 = 8 * C;                // 16-B chunks in the W1c part
    static constexpr int kRowChunks = C / 4;                 // 16-B chunks per W1c row
    static constexpr size_t kLdsBytes = 2 * (size_t)kChunkBytes + 4 * C * 4;   // + bias1
    __device__ static int swz1(int row) { return (C == 96) ? ((row >> 1) & 7) : (row & 15); }
};

template <int C>
__global__ __launch_bounds__(FusedCfg<C>::kThreads) void mlp_fused_kernel(
    const float* __restrict__ y, float* __restrict__ x, const float* __restrict__ wpack /*[chunks][64*C]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M) {
    using Cfg = FusedCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wbuf = smem;                                                   // [2][kChunkBytes]
    float* b1s = reinterpret_cast<float*>(smem + 2 * Cfg::kChunkBytes);  // [4C]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const long long pix0 = (long long)blockIdx.x * Cfg::kPix + wave * 32;
    long long mrow = pix0 + l31;
    const bool valid = mrow < M;
    if (!valid) mrow = M - 1;

    // ---- LDS-DMA source offsets (in floats, relative to the chunk block) for this lane's pieces --------
    int dma_src[Cfg::kPieces];
#pragma unroll
    for (int k = 0; k < Cfg::kPieces; ++k) {
        const int idx = (wave * Cfg::kPieces + k) * 64 + lane;           // linear 16-B slot in the LDS image
        int src;
        if (idx < Cfg::kW1Chunks16) {
            const int row = idx / Cfg::kRowChunks, pos = idx - row * Cfg::kRowChunks;
            src = row * Cfg::kRowChunks + (pos ^ Cfg::swz1(row));
        } else {
            const int i2 = idx - Cfg::kW1Chunks16;
            const int row = i2 >> 3, pos = i2 & 7;
            src = Cfg::kW1Chunks16 + row * 8 + (pos ^ ((row >> 1) & 7));
        }
        dma_src[k] = src * 4;
    }
#define ACX_DMA_CHUNK(j, buf)                                                                                   \
    {                                                                                                           \
        const float* cb = wpack + (long long)(j) * (64 * C);                                                    \
        _Pragma("unroll") for (int k = 0; k < Cfg::kPieces; ++k)                                                \
            __builtin_amdgcn_global_load_lds(                                                                   \
                (const __attribute__((address_space(1))) void*)(cb + dma_src[k]),                               \
                (__attribute__((address_space(3))) void*)(wbuf + (buf) * Cfg::kChunkBytes +                     \
                                                          (wave * Cfg::kPieces + k) * 1024), 16, 0, 0);         \
    }
    ACX_DMA_CHUNK(0, 0);
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

    // fragment read offsets (bytes) in the chunk image
    const int sw1 = Cfg::swz1(l31);
    const int w1row = l31 * (4 * C);
    const int sw2 = (l31 >> 1) & 7;
    const int w2base = Cfg::kW1Chunks16 * 16 + l31 * 128;
    __syncthreads();      // chunk 0 landed (hipcc drains the LDS-DMA in front of the barrier), b1s visible

    for (int j = 0; j < Cfg::kChunks; ++j) {
        const char* wb = wbuf + (j & 1) * Cfg::kChunkBytes;
        if (j + 1 < Cfg::kChunks) ACX_DMA_CHUNK(j + 1, (j + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 1: X^T = W1c . act^T, accumulator pre-loaded with the bias -------------------------
        f32x16 X;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * j + 8 * q + 4 * hh);
            X[4 * q + 0] = bq[0]; X[4 * q + 1] = bq[1]; X[4 * q + 2] = bq[2]; X[4 * q + 3] = bq[3];
        }
#pragma unroll
        for (int t = 0; t < C / 8; ++t) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(wb + w1row + (((hh * (C / 8) + t) ^ sw1) << 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) X = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], act[4 * t + e], X, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) X[r] = gelu_erf_f(X[r]);
        // ---- phase 2: out^T += W2c . X^T --------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < C / 32; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(wb + w2base + t * 4096 + (((2 * q + hh) ^ sw2) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], X[4 * q + e], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#undef ACX_DMA_CHUNK

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
    static bool attr_set = false;
    if (!attr_set) {
        ACX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<C>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::kLdsBytes));
        attr_set = true;
    }
    const long long blocks = (M + Cfg::kPix - 1) / Cfg::kPix;
    mlp_fused_kernel<C><<<dim3((unsigned)blocks), dim3(Cfg::kThreads), Cfg::kLdsBytes, s>>>(y, x, w.wpack, w.b1, w.b2, M);
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
