// K4fs -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86)
// in split-fp16 arithmetic (ACX_PREC_F32_SPLIT, see gemm_split.hip for the number format and its error bound).
// Same dataflow as mlp_fused.hip -- each wave owns 32 pixels, everything is computed transposed so that the
// accumulator tile of the first product is the B operand of the second, the hidden activation never leaves the
// register file, [W1c | W2c] chunk images stream through two LDS rings by LDS-DMA -- but on
// v_mfma_f32_32x32x16_f16 with every fp32 operand carried as fp16 hi + fp16 lo (three MFMAs per product):
//     phase 1   X^T[32 hidden x 32 px] = W1c[32 x C] . LN(y)^T[C x 32 px]     3 C/16 MFMAs; B operand = the wave's
//               normalised activations (x 2^11) as hi/lo halves, resident in C/2 VGPRs: lane (px, h) holds
//               channels 16s + 8h .. +7 of k-step s
//     GELU      on the 16 accumulator registers (bias pre-loaded as the initial accumulator), x 2^h, split into
//               hi/lo halves -- SCALAR fp32 math, 15 VALU per element (split_math.h, gelu_micro2).  The file is built
//               with -fno-slp-vectorize: beside MFMAs a packed-FP32 instruction costs far more than the two scalar
//               ones it replaces (round 2 ran this kernel on v_pk_fma_f32 / v_pk_mul_f32: 104 of them per 36 MFMAs,
//               0.25 of the matrix peak)
//     phase 2   out^T[C x 32 px] += W2c[C x 32 hidden] . G                      3 C/16 MFMAs.  Lane (px, h) holds hidden
//               units 4h + 8q + e (accumulator register 4q + e); registers 8s'..8s'+7 are used as-is for k-step s',
//               i.e. MFMA k-slot (s', h, j) contracts hidden unit 16 s' + 4h + 8 (j>>2) + (j&3) -- W2c is stored
//               with the hidden units of each chunk in exactly that order (acx_finalize), so its fragments are
//               plain 16-B reads.
// HBM traffic per block: read y, read x, write x.
//
// Schedule (per wave, iteration j over the 4C/32 hidden chunks) -- three independent instruction streams:
//     matrix  A: phase 1 of chunk j+2 (chained on one accumulator)       B: phase 2 of chunk j
//     vector  GELU + split of chunk j+1, cut into 64 micro-steps (8 register pairs x 8 steps of 2-6 instructions, two
//             pairs in flight so that consecutive micro-steps are independent) dealt one or two behind EACH MFMA of A
//             and B, behind scheduling fences
//     LDS     fragments of unit u+1 are read while unit u multiplies (two named register pairs)
//     DMA     [W1c | W2c] images run kLead iterations ahead in rings of kLead+1 slots; the end-of-iteration wait is
//             a counted s_waitcnt vmcnt(N) + bare s_barrier so that the youngest images stay in flight (a
//             __syncthreads() would drain them: hipcc puts vmcnt(0) in front of it)
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {



template <int C>
struct FusedSCfg {
    // CU-exclusive workgroups (acx_internal.h): C = 96 needs < 256 registers per lane -> 8 waves (two per SIMD, 512 x 256
    // registers); C = 192 needs more -> 4 waves (one per SIMD, 256 x 512 registers).  Either way the workgroup holds
    // the CU's whole register file and, by request, its whole LDS.
    static constexpr int kWaves = C == 96 ? 8 : 4;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kPix = kWaves * 32;
    // Four loader waves per weight image.  With 8 waves the roles are split: waves 0-3 stream the W1c images, waves
    // 4-7 the W2c images (half the LDS-DMA instructions per wave, and the weight bytes per pixel halve as well).
    static constexpr int kLoaders = 4;
    static constexpr bool kRoleSplit = kWaves == 8;
    static constexpr int kChunks = 4 * C / 32;
    static constexpr int kHalfBytes = 128 * C;               // one [32][C] (or [C][32]) S16 image
    static constexpr int kPieces = kHalfBytes / 1024 / kLoaders;    // 1-KB pieces per loader wave per image
    static constexpr int kRowChunks = C / 4;                 // 16-B chunks per W1c row
    static constexpr int kSteps = C / 16;                    // k-steps of phase 1
    static constexpr int kUnits = 2 * (C / 32);              // (out tile, k-step) units of phase 2
    static constexpr int kLead = 2;                // iterations a weight image is requested ahead of use
    static constexpr int kRing = kLead + 1;
    static constexpr size_t kLdsBytes = 2 * kRing * (size_t)kHalfBytes + 4 * C * 4;
    __device__ static int swz1(int row) { return (C == 96) ? ((row >> 1) & 7) : (row & 15); }
};

template <int C, bool LNOUT>
__global__ __launch_bounds__(FusedSCfg<C>::kThreads) void mlp_fused_split_kernel(
    const float* __restrict__ y, float* __restrict__ x, const char* __restrict__ wpack /*[chunks][256*C bytes]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, float sinv1, float sinv2, float hscale,
    char* __restrict__ ln_out /* LNOUT: (M, C) S16 rows of LayerNorm(x_new) x 2^11, written INSTEAD of x */) {
    using Cfg = FusedSCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1buf = smem;                                                  // [kRing][kHalfBytes]  rows = hidden
    char* w2buf = smem + Cfg::kRing * Cfg::kHalfBytes;                   // [kRing][kHalfBytes]  rows = out channel
    float* b1s = reinterpret_cast<float*>(smem + 2 * Cfg::kRing * Cfg::kHalfBytes);   // [4C], pre-divided by sinv1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    constexpr bool RS = Cfg::kRoleSplit;
    const int lw = RS ? (wave & 3) : wave;           // loader index inside its image
    const int role = RS ? (wave >> 2) : 0;           // RS: 0 = streams W1c images, 1 = streams W2c images
    if constexpr (Cfg::kWaves == 8) { ACX_CLAIM_VGPR(255); } else { ACX_CLAIM_VGPR(255); ACX_CLAIM_AGPR(255); }
    const long long pix0 = (long long)blockIdx.x * Cfg::kPix + wave * 32;
    long long mrow = pix0 + l31;
    const bool valid = mrow < M;
    if (!valid) mrow = M - 1;

    // byte offsets inside a chunk block [W1c | W2c]; with split roles a wave keeps only its own image's (in src1)
    int src1[Cfg::kPieces], src2[RS ? 1 : Cfg::kPieces];
#pragma unroll
    for (int k = 0; k < Cfg::kPieces; ++k) {
        const int idx = (lw * Cfg::kPieces + k) * 64 + lane;           // linear 16-B slot in the LDS image
        const int r1 = idx / Cfg::kRowChunks, p1 = idx - r1 * Cfg::kRowChunks;
        const int o1 = (r1 * Cfg::kRowChunks + (p1 ^ Cfg::swz1(r1))) * 16;
        const int r2 = idx >> 3, p2 = idx & 7;
        const int o2 = Cfg::kHalfBytes + (r2 * 8 + (p2 ^ ((r2 >> 1) & 7))) * 16;
        if (RS) src1[k] = role == 0 ? o1 : o2;
        else { src1[k] = o1; src2[k] = o2; }
    }
    // split roles: this wave's image stream -- chunk lead over the iteration index and ring base
    const int my_lead = Cfg::kLead + (role == 0 ? 2 : 0);
    char* const my_ring = role == 0 ? w1buf : w2buf;
#define ACX_DMA1(srcv, k, j, dstbase)                                                                           \
        __builtin_amdgcn_global_load_lds(                                                                       \
            (const __attribute__((address_space(1))) void*)(wpack + (long long)(j) * (2 * Cfg::kHalfBytes) + srcv[k]), \
            (__attribute__((address_space(3))) void*)((dstbase) + (lw * Cfg::kPieces + (k)) * 1024), 16, 0, 0);
#define ACX_DMA(srcv, j, dstbase)                                                                               \
    {                                                                                                           \
        const char* cb = wpack + (long long)(j) * (2 * Cfg::kHalfBytes);                                        \
        _Pragma("unroll") for (int k = 0; k < Cfg::kPieces; ++k)                                                \
            __builtin_amdgcn_global_load_lds(                                                                   \
                (const __attribute__((address_space(1))) void*)(cb + srcv[k]),                                  \
                (__attribute__((address_space(3))) void*)((dstbase) + (lw * Cfg::kPieces + k) * 1024), 16, 0, 0); \
    }
    constexpr int n = Cfg::kChunks, L = Cfg::kLead, R = Cfg::kRing;
    if constexpr (RS) {         // W1c chunks 0..R-1 (role 0) / W2c chunks 0..L-1 (role 1); W1c chunks R..1+L follow the prologue
#pragma unroll
        for (int c0 = 0; c0 < R; ++c0)
            if (c0 < (role == 0 ? R : L)) ACX_DMA(src1, c0, my_ring + (c0 % R) * Cfg::kHalfBytes);
    } else {
#pragma unroll
        for (int c0 = 0; c0 < R; ++c0) ACX_DMA(src1, c0, w1buf + (c0 % R) * Cfg::kHalfBytes);     // chunks R..1+L follow the prologue
#pragma unroll
        for (int c0 = 0; c0 < L; ++c0) ACX_DMA(src2, c0, w2buf + (c0 % R) * Cfg::kHalfBytes);
    }
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
    const GeluK2 gk = gelu_k2(sinv1, hscale);

#define ACX_H8(v_) __builtin_bit_cast(h8, v_)
#define ACX_FENCE __builtin_amdgcn_sched_barrier(0);
    // phase-1 unit = k-step s_: two fragment reads, three MFMAs chained on Xacc
#define ACX_W1_RD(w1p_, s_, pl_) (*reinterpret_cast<const f32x4*>((w1p_) + w1row + (((2 * (2 * (s_) + hh) + (pl_)) ^ sw1) << 4)))
    // MFMA number m_ of an iteration (kTot of them) is followed, behind scheduling fences, by its share of the 64 GELU
    // micro-steps the iteration carries: steps [64 m / kTot, 64 (m + 1) / kTot)
#define ACX_AFTER(m_) ACX_FENCE if constexpr (HV) { ACX_MICRO(64 * (m_) / kTot, 64 * ((m_) + 1) / kTot) } ACX_FENCE
#define ACX_P1_MFMA(Xacc, s_, ah_, al_, m0_)                                                                    \
        Xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(acth[s_]), Xacc, 0, 0, 0);            \
        ACX_AFTER((m0_) + 0)                                                                                    \
        Xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(actl[s_]), Xacc, 0, 0, 0);            \
        ACX_AFTER((m0_) + 1)                                                                                    \
        Xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(acth[s_]), Xacc, 0, 0, 0);            \
        ACX_AFTER((m0_) + 2)
#define ACX_P1_MFMA_PLAIN(Xacc, s_, ah_, al_)                                                                   \
        Xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(acth[s_]), Xacc, 0, 0, 0);            \
        Xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(actl[s_]), Xacc, 0, 0, 0);            \
        Xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(acth[s_]), Xacc, 0, 0, 0);
    // phase-2 unit i = (tile t = i >> 1, k-step s' = i & 1): fragments of block b = 2s' + hh, hi chunk 2b, lo chunk 2b+1
#define ACX_W2_RD(w2p_, i_, pl_) (*reinterpret_cast<const f32x4*>((w2p_) + ((i_) >> 1) * 4096 + (((2 * (2 * ((i_) & 1) + hh) + (pl_)) ^ sw2) << 4)))
#define ACX_P2_MFMA(i_, ah_, al_, m0_)                                                                          \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(gh[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER((m0_) + 0)                                                                                    \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(gl[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER((m0_) + 1)                                                                                    \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(gh[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER((m0_) + 2)
#define ACX_TOUCH2(h_, l_) { asm volatile("" :: "v"(h_)); asm volatile("" :: "v"(l_)); }
#define ACX_BIAS_INIT(Xacc, j_)                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * (j_) + 8 * q + 4 * hh);                 \
            Xacc[4 * q + 0] = bq[0]; Xacc[4 * q + 1] = bq[1]; Xacc[4 * q + 2] = bq[2]; Xacc[4 * q + 3] = bq[3]; \
        }
    // GELU micro-steps [from, to) of the 64 that turn Xv into the packed halves uh / ul: step m works on register pair
    // 2 (m / 16) + (m & 1) -- two pairs alternate, so neighbouring micro-steps do not depend on each other -- and is
    // step (m % 16) / 2 of that pair's eight
#define ACX_MICRO(from_, to_)                                                                                   \
        _Pragma("unroll") for (int mm_ = (from_); mm_ < (to_); ++mm_) {                                         \
            const int pr_ = 2 * (mm_ / 16) + (mm_ & 1), st_ = (mm_ % 16) >> 1;                                  \
            GeluState2& gs_ = (mm_ & 1) ? gsB : gsA;                                                            \
            if (st_ == 0) { gs_.ax = Xv[2 * pr_]; gs_.ay = Xv[2 * pr_ + 1]; gelu_micro2<0>(gs_, gk, uh[pr_], ul[pr_]); } \
            else if (st_ == 1) gelu_micro2<1>(gs_, gk, uh[pr_], ul[pr_]);                                       \
            else if (st_ == 2) gelu_micro2<2>(gs_, gk, uh[pr_], ul[pr_]);                                       \
            else if (st_ == 3) gelu_micro2<3>(gs_, gk, uh[pr_], ul[pr_]);                                       \
            else if (st_ == 4) gelu_micro2<4>(gs_, gk, uh[pr_], ul[pr_]);                                       \
            else if (st_ == 5) gelu_micro2<5>(gs_, gk, uh[pr_], ul[pr_]);                                       \
            else if (st_ == 6) gelu_micro2<6>(gs_, gk, uh[pr_], ul[pr_]);                                       \
            else gelu_micro2<7>(gs_, gk, uh[pr_], ul[pr_]);                                                     \
        }
#define ACX_PACK_G(dh_, dl_)                                                                                    \
        dh_[0] = __builtin_bit_cast(f32x4, uint4{uh[0], uh[1], uh[2], uh[3]});                                  \
        dh_[1] = __builtin_bit_cast(f32x4, uint4{uh[4], uh[5], uh[6], uh[7]});                                  \
        dl_[0] = __builtin_bit_cast(f32x4, uint4{ul[0], ul[1], ul[2], ul[3]});                                  \
        dl_[1] = __builtin_bit_cast(f32x4, uint4{ul[4], ul[5], ul[6], ul[7]});

    __syncthreads();      // W1c(0..1+L), W2c(0..L-1) landed (hipcc drains the LDS-DMA before the barrier); b1s visible
    f32x16 Xv, Xnn;       // Xv: pre-activation of chunk j+1 (GELU input of iteration j); Xnn: chunk j+2, accumulating
    f32x4 gh[2], gl[2];   // G(j): B operand of phase 2, two k-steps, hi / lo halves
    GeluState2 gsA, gsB;
    unsigned uh[8], ul[8];
    {   // prologue: X(0) -> G(0), X(1) -> Xv; nothing to overlap with yet
        ACX_BIAS_INIT(Xv, 0)
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) {
            const f32x4 ah = ACX_W1_RD(w1buf, s, 0), al = ACX_W1_RD(w1buf, s, 1);
            ACX_P1_MFMA_PLAIN(Xv, s, ah, al)
        }
        ACX_MICRO(0, 64)
        ACX_PACK_G(gh, gl)
        ACX_BIAS_INIT(Xv, 1)
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) {
            const f32x4 ah = ACX_W1_RD(w1buf + Cfg::kHalfBytes, s, 0), al = ACX_W1_RD(w1buf + Cfg::kHalfBytes, s, 1);
            ACX_P1_MFMA_PLAIN(Xv, s, ah, al)
        }
    }
    __syncthreads();      // every wave is done with W1 ring slots 0 and 1 before they are refilled
    if (!RS || role == 0) {
#pragma unroll
        for (int c0 = R; c0 < 2 + L; ++c0) ACX_DMA(src1, c0, w1buf + (c0 % R) * Cfg::kHalfBytes);
    }
    auto body = [&](auto has_a, auto has_v, const int j) __attribute__((always_inline)) {
        constexpr bool HA = decltype(has_a)::value, HV = decltype(has_v)::value;
        constexpr int kRegions = (HA ? Cfg::kSteps : 0) + Cfg::kUnits;
        constexpr int kTot = 3 * kRegions;                     // MFMAs of this iteration
        // this iteration's DMA pieces (W1c(j+2+L) then W2c(j+L)) are threaded through the units, one every
        // kRegions / (2 kPieces) units: issued in a burst each piece costs 100-185 cycles of issue
        constexpr int kPerWave = (RS ? 1 : 2) * Cfg::kPieces;      // pieces a wave issues per iteration
        constexpr int kDmaStride = kRegions / kPerWave;
        static_assert(kDmaStride * kPerWave == kRegions, "DMA pieces must divide over the units");
        const int cjM = j + my_lead;                     // split roles: the chunk this wave requests in this iteration
        const bool dmaM = cjM < n;
        char* const dM = my_ring + (cjM % R) * Cfg::kHalfBytes;
        const bool dma1 = j + 2 + L < n, dma2 = j + L < n;
        char* const d1 = w1buf + ((j + 2 + L) % R) * Cfg::kHalfBytes;
        char* const d2 = w2buf + ((j + L) % R) * Cfg::kHalfBytes;
#define ACX_DMA_AT(region_)                                                                                     \
        if ((region_) % kDmaStride == 0) {                                                                      \
            constexpr int q_ = 0;                                                                               \
            const int qq_ = (region_) / kDmaStride + q_;                                                        \
            if (RS) { if (dmaM) { ACX_DMA1(src1, qq_, cjM, dM) } }                                              \
            else if (qq_ < Cfg::kPieces) { if (dma1) { ACX_DMA1(src1, qq_, j + 2 + L, d1) } }                 \
            else if (dma2) { ACX_DMA1(src2, qq_ - Cfg::kPieces, j + L, d2) }                                    \
        }
        ACX_FENCE
        int region = 0;
        if constexpr (HA) {
            const char* w1p = w1buf + ((j + 2) % R) * Cfg::kHalfBytes;
            ACX_BIAS_INIT(Xnn, j + 2)
            f32x4 a0h = ACX_W1_RD(w1p, 0, 0), a0l = ACX_W1_RD(w1p, 0, 1), a1h, a1l;
#pragma unroll
            for (int s = 0; s < Cfg::kSteps; s += 2) {
                a1h = ACX_W1_RD(w1p, s + 1, 0); a1l = ACX_W1_RD(w1p, s + 1, 1);
                ACX_FENCE
                ACX_P1_MFMA(Xnn, s, a0h, a0l, 3 * (s))
                ACX_DMA_AT(s)
                ACX_FENCE
                ACX_TOUCH2(a1h, a1l)
                if (s + 2 < Cfg::kSteps) { a0h = ACX_W1_RD(w1p, s + 2, 0); a0l = ACX_W1_RD(w1p, s + 2, 1); }
                ACX_FENCE
                ACX_P1_MFMA(Xnn, s + 1, a1h, a1l, 3 * (s + 1))
                ACX_DMA_AT(s + 1)
                ACX_FENCE
                if (s + 2 < Cfg::kSteps) ACX_TOUCH2(a0h, a0l)
            }
            region = Cfg::kSteps;
        }
        {
            const char* w2p = w2buf + (j % R) * Cfg::kHalfBytes + w2row;
            const int r0 = HA ? Cfg::kSteps : 0;
            f32x4 a0h = ACX_W2_RD(w2p, 0, 0), a0l = ACX_W2_RD(w2p, 0, 1), a1h, a1l;
#pragma unroll
            for (int i = 0; i < Cfg::kUnits; i += 2) {
                a1h = ACX_W2_RD(w2p, i + 1, 0); a1l = ACX_W2_RD(w2p, i + 1, 1);
                ACX_FENCE
                ACX_P2_MFMA(i, a0h, a0l, 3 * (r0 + i))
                ACX_DMA_AT(r0 + i)
                ACX_FENCE
                ACX_TOUCH2(a1h, a1l)
                if (i + 2 < Cfg::kUnits) { a0h = ACX_W2_RD(w2p, i + 2, 0); a0l = ACX_W2_RD(w2p, i + 2, 1); }
                ACX_FENCE
                ACX_P2_MFMA(i + 1, a1h, a1l, 3 * (r0 + i + 1))
                ACX_DMA_AT(r0 + i + 1)
                ACX_FENCE
                if (i + 2 < Cfg::kUnits) ACX_TOUCH2(a0h, a0l)
            }
        }
#undef ACX_DMA_AT
        (void)region;
        if constexpr (HV) { ACX_PACK_G(gh, gl) }
        if constexpr (HA) { Xv = Xnn; }
        ACX_FENCE
        // images requested this iteration may stay in flight; everything older must have landed
        if (RS) {
            if (L == 2 && dmaM) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::kPieces) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (L == 2 && j + 2 + L < n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * Cfg::kPieces) : "memory");
        else if (L == 2 && j + L < n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::kPieces) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ACX_FENCE
    };
    for (int j = 0; j + 2 < n; ++j) body(std::true_type{}, std::true_type{}, j);
    body(std::false_type{}, std::true_type{}, n - 2);
    body(std::false_type{}, std::false_type{}, n - 1);
#undef ACX_DMA
#undef ACX_DMA1
#undef ACX_H8
#undef ACX_FENCE
#undef ACX_W1_RD
#undef ACX_P1_MFMA
#undef ACX_P1_MFMA_PLAIN
#undef ACX_AFTER
#undef ACX_W2_RD
#undef ACX_P2_MFMA
#undef ACX_TOUCH2
#undef ACX_BIAS_INIT
#undef ACX_MICRO
#undef ACX_PACK_G

    // ---- epilogue: lane (px, hh), tile t, q: channels 32t + 8q + 4hh .. +3  ->  x = x + out + b2 ---------
    if constexpr (LNOUT) {
        // Last block of a stage in the full forward: the only reader of the new x is the LayerNorm in front of the
        // downsample conv (convnext.py:230-235 -- no skip connection leaves the stage), and this wave holds whole
        // rows (a lane and its partner at lane ^ 32 hold all C channels of one pixel).  So normalise here and
        // write the S16 operand of the downsample GEMM (same form as rowstats_kernel<.,3>: two-pass statistics,
        // biased variance, eps 1e-6, x 2^11, blocks [8 hi][8 lo]) in place of x: one tensor pass less on each
        // side and no LayerNorm launch.  ln_out may alias y: a workgroup reads its own rows of y at the start only.
        const float* xp = x + mrow * C + 4 * hh;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < C / 32; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * t + 8 * q;
                const float4 bb = *reinterpret_cast<const float4*>(b2 + c + 4 * hh);
                const float4 v = *reinterpret_cast<const float4*>(xp + c);
                acc[t][4 * q + 0] = v.x + fmaf(acc[t][4 * q + 0], sinv2, bb.x);
                acc[t][4 * q + 1] = v.y + fmaf(acc[t][4 * q + 1], sinv2, bb.y);
                acc[t][4 * q + 2] = v.z + fmaf(acc[t][4 * q + 2], sinv2, bb.z);
                acc[t][4 * q + 3] = v.w + fmaf(acc[t][4 * q + 3], sinv2, bb.w);
                sum += (acc[t][4 * q + 0] + acc[t][4 * q + 1]) + (acc[t][4 * q + 2] + acc[t][4 * q + 3]);
            }
        }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int t = 0; t < C / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float u = acc[t][r] - mean; d = fmaf(u, u, d); }
        d += __shfl_xor(d, 32);
        const float sc = kSplitLnScale / sqrtf(d * (1.0f / C) + 1e-6f);
        if (valid) {
            char* op = ln_out + mrow * (long long)(C * 4) + 8 * hh;
#pragma unroll
            for (int t = 0; t < C / 32; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned uhi[2], ulo[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        f32x2 v;
                        v.x = (acc[t][4 * q + 2 * e] - mean) * sc; v.y = (acc[t][4 * q + 2 * e + 1] - mean) * sc;
                        const h2 h = __builtin_convertvector(v, h2);
                        const f32x2 back = __builtin_convertvector(h, f32x2);
                        const h2 l = __builtin_convertvector(v - back, h2);
                        uhi[e] = __builtin_bit_cast(unsigned, h);
                        ulo[e] = __builtin_bit_cast(unsigned, l);
                    }
                    char* blk = op + (4 * t + q) * 32;          // channels 32t + 8q .. +7: this lane the half 4hh .. +3
                    *reinterpret_cast<uint2*>(blk) = uint2{uhi[0], uhi[1]};
                    *reinterpret_cast<uint2*>(blk + 16) = uint2{ulo[0], ulo[1]};
                }
            }
        }
    } else if (valid) {
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

template <int C, bool LNOUT>
static int launch_fused_s_cfg(const BlockW& w, const float* y, float* x, long long M, void* ln_out, hipStream_t s) {
    using Cfg = FusedSCfg<C>;
    static DeviceOnce once;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "weight rings do not fit the LDS");
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_split_kernel<C, LNOUT>, kCuLdsBytes));
    const long long blocks = (M + Cfg::kPix - 1) / Cfg::kPix;
    mlp_fused_split_kernel<C, LNOUT><<<dim3((unsigned)blocks), dim3(Cfg::kThreads), kCuLdsBytes /* all of it: CU-exclusive */, s>>>(
        y, x, reinterpret_cast<const char*>(w.wpack_s), w.b1, w.b2, M, 1.0f / (kSplitLnScale * w.w1s_scale),
        1.0f / (w.hid_scale * w.w2s_scale), w.hid_scale, reinterpret_cast<char*>(ln_out));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

bool mlp_fused_split_supported(int C) { return C == 96; }     // (C = 192, 384: mlp_fused_wide.hip)

int launch_mlp_fused_split(acx_ctx* c, const BlockW& w, int C, const float* y, float* x, long long M, hipStream_t s,
                           void* ln_out) {
    if (!w.wpack_s) ACX_FAIL(ACX_ERR_STATE, "fused split MLP: chunk-major S16 weights were not packed for C=%d", C);
    ProfScope ps(c, ACX_K_MLP_FUSED, s);
    if (C == 96) return ln_out ? launch_fused_s_cfg<96, true>(w, y, x, M, ln_out, s) : launch_fused_s_cfg<96, false>(w, y, x, M, nullptr, s);
    ACX_FAIL(ACX_ERR_SHAPE, "fused split MLP: unsupported channel count %d", C);
}

}  // namespace acx
