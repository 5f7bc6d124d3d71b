// K4/K5 -- the dense contractions of the backbone on the f32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products, k-ordered fp32 accumulation):
//   pwconv1 + GELU   (Block.pwconv1 / act,  convnext.py:62-65, :79-80)   A = LayerNorm(y)
//   pwconv2 + gamma + residual (convnext.py:66-71, :81-86)               A = hidden
//   downsample 2x2/s2 conv as a GEMM over K = 4C (convnext.py:230-235)   A = LayerNorm(x) gathered
// out[M,N] = epi( A'[M,K] . Wt[N,K]^T + bias[N] ),  A' = (A - mean_row) * rstd_row when stats are
// given (the LayerNorm affine is folded into Wt/bias at acx_finalize).
//
// Tiling: 128 x BN x 32 per workgroup, 4 waves; each wave owns TM x TN tiles of 32x32 and steps K by
// 2 per MFMA.  Operand fragments are read from LDS as one ds_read_b128 per 4 k-steps: lane half h
// takes floats [8g+4h, 8g+4h+4) of its row, so MFMA j of group g contracts k = {8g+j, 8g+4+j};
// A and B use the same permutation, which leaves the sum unchanged.  LDS rows are padded to 36
// floats: the 16-lane groups of ds_read_b128 then hit 16 distinct 4-bank slots (conflict-free).
// Global->LDS staging goes through registers (next tile's loads are issued before the MFMAs of the
// current one) because the A path applies LayerNorm / the 2x2 gather on the way.
#include "acx_internal.h"

namespace acx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 128;
constexpr int kBK = 32;
constexpr int kLdsStride = kBK + 4;

__device__ __forceinline__ float erf_as(float x) {
    // Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 (fp32 erf itself carries ~1e-7)
    const float ax = fabsf(x);
    const float t = 1.0f / fmaf(0.3275911f, ax, 1.0f);
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);
    const float y = fmaf(-p * t, e, 1.0f);
    return copysignf(y, x);
}

__device__ __forceinline__ float gelu_erf(float x) {   // nn.GELU() default (approximate='none')
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}

struct GemmParams {
    const float* A; const float* Wt; const float* bias; float* out; const float* stats; const float* resid;
    long long M; int N; int K;
    int H, W, C, Ho, Wo;      // gather mode
    int tiles_n;
};

// AMODE: 0 plain rows, 1 rows + LayerNorm, 2 2x2 gather + LayerNorm
//
// Pipeline per 32-deep k-tile (LDS double-buffered, ONE barrier per tile):
//   issue the global loads of tile t+1 into registers (raw)      -- latency hides under the MFMAs
//   64 MFMAs per wave on tile t from LDS buffer t&1
//   LayerNorm the staged registers, write them to LDS buffer (t+1)&1, barrier
template <int BN, int WM, int WN, int EPI, int AMODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int TM = kBM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int A_F4 = kBM * kBK / 4 / 256;     // float4 per thread per tile (4)
    constexpr int B_F4 = BN * kBK / 4 / 256;      // 4 (BN=128) or 3 (BN=96)
    constexpr int A_TILE = kBM * kLdsStride, B_TILE = BN * kLdsStride;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* As = smem_f;                  // [2][A_TILE]
    float* Bs = smem_f + 2 * A_TILE;     // [2][B_TILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_n = blockIdx.x % p.tiles_n;
    const long long tile_m = blockIdx.x / p.tiles_n;
    const long long m0 = tile_m * kBM;
    const int n0 = tile_n * BN;

    // ---- per-thread staging coordinates -------------------------------------------------------
    const int c4 = tid & 7;           // which float4 of the 32-float k-slab
    const int r0 = tid >> 3;          // row within a 32-row group
    const float* a_ptr[A_F4];
    float2 a_st[A_F4];                // (mean, rstd) of the row (AMODE 1)
    long long a_pix[A_F4];            // AMODE 2: top-left input pixel of the 2x2 patch
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        long long m = m0 + r0 + 32 * i;
        if (m >= p.M) m = p.M - 1;
        a_st[i] = make_float2(0.f, 1.f);
        a_pix[i] = 0;
        a_ptr[i] = p.A;
        if (AMODE == 2) {
            const int wo = (int)(m % p.Wo);
            const long long t = m / p.Wo;
            const int ho = (int)(t % p.Ho);
            const long long b = t / p.Ho;
            a_pix[i] = (b * p.H + 2 * ho) * p.W + 2 * wo;
        } else {
            a_ptr[i] = p.A + m * p.K + 4 * c4;
            if (AMODE == 1) a_st[i] = *reinterpret_cast<const float2*>(p.stats + 2 * m);
        }
    }
    const float* b_ptr = p.Wt + (long long)(n0 + r0) * p.K + 4 * c4;
    const long long b_step = 32LL * p.K;

    // Staged tile registers are NAMED scalars on purpose: as arrays hipcc (ROCm 7.2) leaves them in
    // scratch memory (un-promoted alloca) once a sched_barrier sits between their def and use.
    f32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    float2 rs0, rs1, rs2, rs3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = f32x4{0.f, 0.f, 0.f, 0.f};
    rs0 = rs1 = rs2 = rs3 = make_float2(0.f, 1.f);
#define ACX_LOAD_A(i, k0)                                                                              \
    if (AMODE == 2) {                                                                                  \
        const int qd = (k0) / p.C;                                                                     \
        const int cc = (k0) - qd * p.C;                                                                \
        const long long pix = a_pix[i] + (long long)(qd >> 1) * p.W + (qd & 1);                        \
        ra##i = *reinterpret_cast<const f32x4*>(p.A + pix * p.C + cc + 4 * c4);                       \
        rs##i = *reinterpret_cast<const float2*>(p.stats + 2 * pix);                                   \
    } else {                                                                                           \
        ra##i = *reinterpret_cast<const f32x4*>(a_ptr[i] + (k0));                                     \
        rs##i = a_st[i];                                                                               \
    }
#define ACX_LOAD_B(i, k0) \
    if (i < B_F4) rb##i = *reinterpret_cast<const f32x4*>(b_ptr + i * b_step + (k0));
#define ACX_LOAD_TILE(k0)                                                                              \
    {                                                                                                  \
        ACX_LOAD_A(0, k0) ACX_LOAD_A(1, k0) ACX_LOAD_A(2, k0) ACX_LOAD_A(3, k0)                        \
        ACX_LOAD_B(0, k0) ACX_LOAD_B(1, k0) ACX_LOAD_B(2, k0) ACX_LOAD_B(3, k0)                        \
    }
#define ACX_STORE_A(i, as_)                                                                            \
    {                                                                                                  \
        f32x4 v = ra##i;                                                                               \
        if (AMODE != 0) v = (v - rs##i.x) * rs##i.y;                                                   \
        *reinterpret_cast<f32x4*>(&(as_)[(r0 + 32 * i) * kLdsStride + 4 * c4]) = v;                    \
    }
#define ACX_STORE_B(i, bs_) \
    if (i < B_F4) *reinterpret_cast<f32x4*>(&(bs_)[(r0 + 32 * i) * kLdsStride + 4 * c4]) = rb##i;
#define ACX_STORE_TILE(buf)                                                                            \
    {                                                                                                  \
        float* as_ = As + (buf) * A_TILE;                                                              \
        float* bs_ = Bs + (buf) * B_TILE;                                                              \
        ACX_STORE_A(0, as_) ACX_STORE_A(1, as_) ACX_STORE_A(2, as_) ACX_STORE_A(3, as_)                \
        ACX_STORE_B(0, bs_) ACX_STORE_B(1, bs_) ACX_STORE_B(2, bs_) ACX_STORE_B(3, bs_)                \
    }
    static_assert(A_F4 == 4 && B_F4 <= 4, "staging macros assume 4 A rows and <= 4 B rows per thread");

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / kBK;
    ACX_LOAD_TILE(0);
    ACX_STORE_TILE(0);
    __syncthreads();
    const int a_frag_off = (wm * TM * 32 + l31) * kLdsStride + 4 * hh;
    const int b_frag_off = (wn * TN * 32 + l31) * kLdsStride + 4 * hh;
#define ACX_COMPUTE_TILE(buf)                                                                          \
    {                                                                                                  \
        const float* a_frag_base = As + (buf) * A_TILE + a_frag_off;                                   \
        const float* b_frag_base = Bs + (buf) * B_TILE + b_frag_off;                                   \
        _Pragma("unroll") for (int g = 0; g < kBK / 8; ++g) {                                          \
            float4 af[TM], bf[TN];                                                                     \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
                af[i] = *reinterpret_cast<const float4*>(a_frag_base + i * 32 * kLdsStride + 8 * g);   \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                             \
                bf[j] = *reinterpret_cast<const float4*>(b_frag_base + j * 32 * kLdsStride + 8 * g);   \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0); \
            }                                                                                          \
        }                                                                                              \
    }
    for (int kt = 0; kt + 1 < nk; ++kt) {
        ACX_LOAD_TILE((kt + 1) * kBK);
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE the MFMAs (hipcc sinks it otherwise)
        ACX_COMPUTE_TILE(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        // opaque re-definition: pins the LayerNorm math and the vmcnt wait BELOW the MFMA block
        asm volatile("" : "+v"(ra0), "+v"(ra1), "+v"(ra2), "+v"(ra3), "+v"(rb0), "+v"(rb1), "+v"(rb2), "+v"(rb3));
        ACX_STORE_TILE((kt + 1) & 1);
        __syncthreads();
    }
    ACX_COMPUTE_TILE((nk - 1) & 1);
#undef ACX_COMPUTE_TILE
#undef ACX_LOAD_TILE
#undef ACX_STORE_TILE
#undef ACX_LOAD_A
#undef ACX_LOAD_B
#undef ACX_STORE_A
#undef ACX_STORE_B

    // ---- epilogue: D tile layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m) -----
    const bool full = m0 + kBM <= p.M;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        const float bn = p.bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long mb = m0 + (wm * TM + i) * 32 + 4 * hh;
            float* op = p.out + mb * p.N + n;
            const float* rp = (EPI == EPI_RESID) ? p.resid + mb * p.N + n : nullptr;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long off = (long long)((r & 3) + 8 * (r >> 2)) * p.N;
                    float v = acc[i][j][r] + bn;
                    if (EPI == EPI_GELU) v = gelu_erf(v);
                    if (EPI == EPI_RESID) v += rp[off];
                    op[off] = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (mb + dr < p.M) {
                        float v = acc[i][j][r] + bn;
                        if (EPI == EPI_GELU) v = gelu_erf(v);
                        if (EPI == EPI_RESID) v += rp[(long long)dr * p.N];
                        op[(long long)dr * p.N] = v;
                    }
                }
            }
        }
    }
}

template <int BN>
constexpr size_t gemm_lds_bytes() { return (size_t)2 * (kBM + BN) * kLdsStride * sizeof(float); }

template <int BN, int WM, int WN, int EPI, int AMODE>
static int launch_cfg(const GemmParams& p0, hipStream_t s) {
    GemmParams p = p0;
    p.tiles_n = p.N / BN;
    const long long tiles_m = (p.M + kBM - 1) / kBM;
    const long long blocks = tiles_m * p.tiles_n;
    if (blocks > 0x7fffffffLL) ACX_FAIL(ACX_ERR_SHAPE, "gemm: grid too large");
    static bool attr_set = false;
    if (!attr_set) {
        ACX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BN, WM, WN, EPI, AMODE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes<BN>()));
        attr_set = true;
    }
    gemm_f32_kernel<BN, WM, WN, EPI, AMODE><<<dim3((unsigned)blocks), dim3(256), gemm_lds_bytes<BN>(), s>>>(p);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

template <int EPI, int AMODE>
static int launch_bn(const GemmParams& p, hipStream_t s) {
    if (p.N % 128 == 0) return launch_cfg<128, 2, 2, EPI, AMODE>(p, s);
    if (p.N % 96 == 0) return launch_cfg<96, 4, 1, EPI, AMODE>(p, s);
    ACX_FAIL(ACX_ERR_SHAPE, "gemm: N=%d is not a multiple of 96 or 128", p.N);
}

int launch_gemm(acx_ctx* c, const GemmArgs& a, hipStream_t s) {
    if (a.K % kBK != 0) ACX_FAIL(ACX_ERR_SHAPE, "gemm: K=%d is not a multiple of %d", a.K, kBK);
    if (a.M <= 0) return ACX_OK;
    GemmParams p;
    p.A = a.A; p.Wt = a.Wt; p.bias = a.bias; p.out = a.out; p.stats = a.stats; p.resid = a.resid;
    p.M = a.M; p.N = a.N; p.K = a.K; p.H = a.H; p.W = a.W; p.C = a.C; p.Ho = a.Ho; p.Wo = a.Wo;
    p.tiles_n = 0;
    ProfScope ps(c, a.cls, s);
    if (a.gather) {
        if (!a.stats || a.epi != EPI_BIAS || a.C % kBK != 0) ACX_FAIL(ACX_ERR_ARG, "gemm: bad gather configuration");
        return launch_bn<EPI_BIAS, 2>(p, s);
    }
    if (a.epi == EPI_GELU && a.stats) return launch_bn<EPI_GELU, 1>(p, s);
    if (a.epi == EPI_RESID && !a.stats) return launch_bn<EPI_RESID, 0>(p, s);
    if (a.epi == EPI_BIAS && !a.stats) return launch_bn<EPI_BIAS, 0>(p, s);
    ACX_FAIL(ACX_ERR_ARG, "gemm: unsupported epilogue/prologue combination (epi=%d, stats=%d)", a.epi,
             a.stats != nullptr);
}

}  // namespace acx
