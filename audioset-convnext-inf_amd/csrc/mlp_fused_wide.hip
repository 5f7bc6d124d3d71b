// K4fw -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86) in
// split-fp16 arithmetic for WIDE stages (C = 384): the hidden activation (B H W x 4C, 347 MB per block at B = 64) never
// exists in memory (hard part 1).  Same transposed dataflow as mlp_fused_split.hip -- a wave owns 32 pixels,
// the accumulator tile of the first product is the B operand of the second -- re-planned for one wave per SIMD:
//
//   registers   the workgroup is CU-exclusive (acx_internal.h): 4 waves x 512 registers.  Per lane: C/2 for the wave's
//               normalised activations (hi + lo halves, B operand of phase 1), C/2 accumulators of out^T, ~100 working.
//   weights     acx_finalize lays every block's weights out as ONE stream of segments in consumption order,
//                 W1(0) W1(1) W2(0) W1(2) W2(1) ... W1(n-1) W2(n-2) W2(n-1)        (n = 4C/32 hidden chunks)
//               a segment = the [32 x C] (W1) or [C x 32] (W2) S16 image of one chunk, 128 C bytes, already in LDS image
//               order (XOR swizzle baked in), so the LDS-DMA source of a piece is base + 1 KB x piece + 16 x lane.
//               LDS holds a ring of three segments: one being multiplied, one landed or landing, one being requested.
//   schedule    segment 2k-1: phase 1 of chunk k    X^T[32 hidden x 32 px] = W1c . LN(y)^T     3 C/16 MFMAs on Xn
//               segment 2k  : phase 2 of chunk k-1  out^T[C x 32 px] += W2c . G(k-1)            3 C/16 MFMAs
//               GELU + hi/lo split of X(k) -> G(k): 30 single-instruction "nano-steps" per register pair (split_math.h,
//               gelu_nano), half of the pairs dealt over the MFMAs of segment 2k, the other half over those of segment
//               2k+1.  One wave per SIMD issues both streams: about five single-issue instructions ride for free behind
//               a 32x32x16 MFMA, every further one costs ~5 cycles (profiles/r03_d_coissue_table_full.txt), so each MFMA
//               gap gets at most four slots, fewer where an LDS-DMA piece or the fragment reads of the next unit already
//               sit in it (WideCfg::gap_free).
//               The DMA pieces of segment s+2 are threaded through the units of segment s; one counted s_waitcnt vmcnt +
//               s_barrier per segment.
// HBM traffic per block: read y, read x, write x (3 C H W 4 bytes) + the weight stream from L2 / MALL.
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {

// Diagnostic builds only (-DACX_FW_STAMPS; tools/lab): s_memtime differences per section, summed per wave.  The product
// build contains no stamp.
#ifdef ACX_FW_STAMPS
__device__ unsigned long long acx_fw_stamps[2048 * 4 * 8];
#define ACX_WSTAMP_DECL unsigned long long st_prev_ = 0, st_sum_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ACX_WSTAMP(k_)                                                                                          \
    {                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        unsigned long long t_;                                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                            \
        if ((k_) >= 0) st_sum_[(k_) < 0 ? 0 : (k_)] += t_ - st_prev_;                                           \
        st_prev_ = t_;                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    }
#define ACX_WSTAMP_FLUSH                                                                                        \
    if (lane == 0 && blockIdx.x < 2048) { _Pragma("unroll") for (int k = 0; k < 8; ++k) acx_fw_stamps[(blockIdx.x * 4 + wave) * 8 + k] = st_sum_[k]; }
#else
#define ACX_WSTAMP_DECL
#define ACX_WSTAMP(k_)
#define ACX_WSTAMP_FLUSH
#endif

// PT = pixel tiles (of 32) per wave: every weight fragment read from LDS feeds PT x 3 MFMAs.  Shipped with PT = 1.
// Measured for C = 96 (stage 0), where a segment is only 18 MFMAs per pixel tile against the fixed cost of a segment
// (three LDS-DMA issues, the barrier) and 36 GELU micro-steps: PT = 2 is correct (parity suite green) but runs 647 us per
// block against 595 for the 8-wave mlp_fused_split_kernel<96>, which therefore keeps stage 0.
template <int C, int PT>
struct WideCfg {
    static constexpr int kWaves = 4;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kPix = kWaves * 32 * PT;
    static constexpr int kChunks = 4 * C / 32;              // n
    static constexpr int kSegs = 2 * kChunks;
    static constexpr int kSegBytes = 128 * C;               // one [32][C] or [C][32] S16 image
    static constexpr int kPieces = kSegBytes / 1024 / kWaves;   // 1-KB LDS-DMA pieces per wave per segment
    static constexpr int kSteps = C / 16;                   // units of a phase-1 segment (k-steps)
    static constexpr int kUnits = 2 * (C / 32);             // units of a phase-2 segment (out tile, k-step)
    static constexpr size_t kLdsBytes = 3 * (size_t)kSegBytes + 4 * C * 4;
    static constexpr int kMfmas = 3 * kUnits * PT;          // MFMAs per segment
    static constexpr int kDmaStride = kUnits / kPieces;     // one LDS-DMA piece every kDmaStride units
    // GELU nano-steps (single instructions, split_math.h) a segment carries: half of a pixel tile's 8 register pairs
    static constexpr int kNano = 4 * kGeluNano * PT;
    static_assert(C % 32 == 0 && kSteps == kUnits && kUnits % kPieces == 0, "unit / piece bookkeeping");
    // Issue budget of the gap BEHIND MFMA m of a segment (unit m / 3 / PT, position m % 3 for PT = 1).  A gap hides about
    // five single-issue instructions (profiles/r03_a_coissue_table.txt); the fixed tenants are the LDS-DMA piece (s_mov m0,
    // s_nop, the load: behind the FIRST MFMA of every kDmaStride-th unit) and the counted wait + two fragment reads (behind
    // the LAST MFMA of every unit).  What is left of four slots per gap is dealt to the GELU in proportion.
    // (closed forms, no loops: the arguments become constants only after the unit loops are unrolled, and everything derived
    // from them -- register indices above all -- must fold then)
    __host__ __device__ static constexpr int cum_free_units(int u) { return 9 * u - 3 * ((u + kDmaStride - 1) / kDmaStride); }   // gaps of units [0, u)
    __host__ __device__ static constexpr int cum_free(int m) {       // free slots of gaps [0, m]
        const int g = m / PT, u = g / 3, pos = g % 3;
        const int f0 = (u % kDmaStride == 0) ? 1 : 4;
        return PT * (cum_free_units(u) + (pos == 0 ? f0 : (pos == 1 ? f0 + 4 : f0 + 5)));
    }
    __host__ __device__ static constexpr int nano_end(int m) {      // nano-steps issued once the gap behind MFMA m is done
        constexpr int tot = PT * cum_free_units(kUnits);
        return (kNano * cum_free(m) + tot / 2) / tot;
    }
    __host__ __device__ static constexpr int nano_begin(int m) { return m == 0 ? 0 : nano_end(m - 1); }
    // W1 rows are 4 C bytes: the XOR that spreads 16 consecutive rows over the LDS banks (api.hip packs the image with it)
    __device__ static int swz1(int row) { return (C % 64 == 0) ? (row & 15) : ((row >> 1) & 7); }
};

// byte offset of segment s in the stream: order W1(0) W1(1) W2(0) W1(2) W2(1) ... (see the header)
//   W1(k): k == 0 ? 0 : 2k - 1       W2(k): k == n - 1 ? 2n - 1 : 2k + 2
template <int C, int PT, bool LNOUT>
__global__ __launch_bounds__(256) void mlp_fused_wide_kernel(
    const float* __restrict__ y, float* __restrict__ x, const char* __restrict__ wstream /*[2n][128 C bytes]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, float sinv1, float sinv2, float hscale,
    char* __restrict__ ln_out /* LNOUT: (M, C) S16 rows of LayerNorm(x_new) x 2^11, written INSTEAD of x */) {
    using Cfg = WideCfg<C, PT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + 3 * Cfg::kSegBytes);   // [4C], pre-divided by sinv1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    ACX_CLAIM_VGPR(255);          // CU-exclusive: one wave per SIMD holds the SIMD's whole register file
    ACX_CLAIM_AGPR(255);
    ACX_WSTAMP_DECL
    ACX_WSTAMP(-1)
    long long mrow[PT];           // this lane's pixel row in each of the wave's pixel tiles
    bool valid[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        mrow[pt] = (long long)blockIdx.x * Cfg::kPix + (wave * PT + pt) * 32 + l31;
        valid[pt] = mrow[pt] < M;
        if (!valid[pt]) mrow[pt] = M - 1;
    }

    constexpr int n = Cfg::kChunks;
    const int dma_lane = (wave * Cfg::kPieces) * 1024 + lane * 16;      // this lane's slot in piece 0 of its wave
    // (issued from inline asm: next to the builtin form hipcc waits lgkmcnt(0) / vmcnt(0) wherever an LDS read follows, which
    // defeats the two-unit look-ahead of the fragment reads; the counted waits at the segment ends are this file's own)
    const unsigned smem_a = acx_lds_addr(smem);
#define ACX_WDMA(seg_, piece_, grp_)                                                                             \
        acx_glds16_own_m0(wstream + (long long)(seg_) * Cfg::kSegBytes + dma_lane + (piece_) * 1024,             \
                          __builtin_amdgcn_readfirstlane(smem_a + (grp_) * Cfg::kSegBytes + (wave * Cfg::kPieces + (piece_)) * 1024));
    // segments 0 and 1 are requested before anything else; segment s + 2 follows during segment s
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(0, p, 0)
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(1, p, 1)
    {
        const float b1scale = 1.0f / sinv1;             // a power of two
        for (int i = tid; i < 4 * C; i += Cfg::kThreads) b1s[i] = b1[i] * b1scale;
    }

    // ---- this wave's activations: lane (px = l31, half hh) holds channels 16s + 8hh .. +7, s = 0..C/16-1 --------
    f32x4 acth[PT][Cfg::kSteps], actl[PT][Cfg::kSteps];         // 8 fp16 halves each
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        float a[C / 2];
        const float* yp = y + mrow[pt] * C + 8 * hh;
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
            unsigned uh4[4], ul4[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x2 v;
                v.x = (a[8 * s + 2 * p] - mean) * sc; v.y = (a[8 * s + 2 * p + 1] - mean) * sc;
                const h2 h = __builtin_convertvector(v, h2);
                const f32x2 back = __builtin_convertvector(h, f32x2);
                const h2 l = __builtin_convertvector(v - back, h2);
                uh4[p] = __builtin_bit_cast(unsigned, h);
                ul4[p] = __builtin_bit_cast(unsigned, l);
            }
            acth[pt][s] = __builtin_bit_cast(f32x4, uint4{uh4[0], uh4[1], uh4[2], uh4[3]});
            actl[pt][s] = __builtin_bit_cast(f32x4, uint4{ul4[0], ul4[1], ul4[2], ul4[3]});
        }
    }

    f32x16 acc[PT][C / 32];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int t = 0; t < C / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pt][t][r] = 0.f;

    // fragment addresses inside a segment (without the ring offset):
    //   W1 image: row = hidden unit l31 (4 C bytes = C/4 chunks of 16 B), chunk p = 4 s + 2 hh + pl at position p ^ swz1(l31)
    //             (pl = 0 hi halves, 1 lo halves; the XOR touches the low 4 (C = 96: 3) bits only)
    //   W2 image: row = out channel (128 B = 8 chunks), tile t rows 32 t + l31, chunk 2 (2 s' + hh) + pl at position ^ ((l31 >> 1) & 7)
    int w1off[4][2], w2off[2][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w1off[q][pl] = l31 * (4 * C) + (((4 * q + 2 * hh + pl) ^ Cfg::swz1(l31)) << 4);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w2off[sp][pl] = l31 * 128 + (((2 * (2 * sp + hh) + pl) ^ ((l31 >> 1) & 7)) << 4);
    const GeluK2 gk = gelu_k2(sinv1, hscale);

#define ACX_H8(v_) __builtin_bit_cast(h8, v_)
#define ACX_FENCE __builtin_amdgcn_sched_barrier(0);
    // phase-1 unit = k-step s_ of the chunk: chunk 4 s_ + ..: the bits above the XORed four = s_ / 4 -> + 256 B each
#define ACX_W1_RD(base_, s_, pl_) (*reinterpret_cast<const f32x4*>((base_) + ((s_) >> 2) * 256 + w1off[(s_) & 3][pl_]))
    // MFMA number m_ of a segment is followed (behind a scheduling fence) by its share of the kHalf GELU micro-steps the
    // segment carries: steps [kHalf m / kMfmas, kHalf (m + 1) / kMfmas) of the half (C = 384: one after every other MFMA,
    // C = 192: one after each, C = 96: two after each)
#define ACX_AFTER_MFMA(HV_, half_, m_)                                                                          \
        ACX_FENCE if constexpr (HV_) { ACX_NANO_RANGE(half_, Cfg::nano_begin(m_), Cfg::nano_end(m_)) } ACX_FENCE
    // the three terms of a unit, each over the PT pixel tiles (independent accumulators back to back)
#define ACX_P1_MFMA(s_, ah_, al_)                                                                               \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            Xn[pt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(acth[pt_][s_]), Xn[pt_], 0, 0, 0); \
            if (pt_ == 0 && (s_) % Cfg::kDmaStride == 0 && dma) { ACX_WDMA(seg_ + 2, (s_) / Cfg::kDmaStride, g2) } \
            ACX_AFTER_MFMA(HV, 1, (3 * (s_) + 0) * PT + pt_) }                                                  \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            Xn[pt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(actl[pt_][s_]), Xn[pt_], 0, 0, 0); \
            ACX_AFTER_MFMA(HV, 1, (3 * (s_) + 1) * PT + pt_) }                                                  \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            Xn[pt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(acth[pt_][s_]), Xn[pt_], 0, 0, 0); \
            ACX_AFTER_MFMA(HV, 1, (3 * (s_) + 2) * PT + pt_) }
    // phase-2 unit i = (out tile t = i >> 1, k-step s' = i & 1)
#define ACX_W2_RD(base_, i_, pl_) (*reinterpret_cast<const f32x4*>((base_) + ((i_) >> 1) * 4096 + w2off[(i_) & 1][pl_]))
#define ACX_P2_MFMA(i_, ah_, al_)                                                                               \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            acc[pt_][(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(gh[pt_][(i_) & 1]), acc[pt_][(i_) >> 1], 0, 0, 0); \
            if (pt_ == 0 && (i_) % Cfg::kDmaStride == 0 && dma) { ACX_WDMA(seg_ + 2, (i_) / Cfg::kDmaStride, g2) } \
            ACX_AFTER_MFMA(HV, 0, (3 * (i_) + 0) * PT + pt_) }                                                  \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            acc[pt_][(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(gl[pt_][(i_) & 1]), acc[pt_][(i_) >> 1], 0, 0, 0); \
            ACX_AFTER_MFMA(HV, 0, (3 * (i_) + 1) * PT + pt_) }                                                  \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            acc[pt_][(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(gh[pt_][(i_) & 1]), acc[pt_][(i_) >> 1], 0, 0, 0); \
            ACX_AFTER_MFMA(HV, 0, (3 * (i_) + 2) * PT + pt_) }
    // nano-steps [from, to) of the kNano that segment half half_ carries: step ng_ belongs to pixel tile ng_ / 120 and is
    // instruction ng_ % 30 of register pair 4 half_ + (ng_ % 120) / 30 (split_math.h, gelu_nano): pair after pair, in order
#define ACX_NANO_RANGE(half_, from_, to_)                                                                       \
        _Pragma("unroll") for (int ng_ = (from_); ng_ < (to_); ++ng_) {                                         \
            const int mt_ = ng_ / (4 * kGeluNano), pr_ = 4 * (half_) + (ng_ % (4 * kGeluNano)) / kGeluNano, st_ = ng_ % kGeluNano; \
            if (st_ == 0) { gsA.ax = Xv[mt_][2 * pr_]; gsA.ay = Xv[mt_][2 * pr_ + 1]; gelu_nano<0>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]); } \
            else if (st_ == 1) gelu_nano<1>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 2) gelu_nano<2>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 3) gelu_nano<3>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 4) gelu_nano<4>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 5) gelu_nano<5>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 6) gelu_nano<6>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 7) gelu_nano<7>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 8) gelu_nano<8>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 9) gelu_nano<9>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 10) gelu_nano<10>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 11) gelu_nano<11>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 12) gelu_nano<12>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 13) gelu_nano<13>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 14) gelu_nano<14>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 15) gelu_nano<15>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 16) gelu_nano<16>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 17) gelu_nano<17>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 18) gelu_nano<18>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 19) gelu_nano<19>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 20) gelu_nano<20>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 21) gelu_nano<21>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 22) gelu_nano<22>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 23) gelu_nano<23>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 24) gelu_nano<24>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 25) gelu_nano<25>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 26) gelu_nano<26>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 27) gelu_nano<27>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 28) gelu_nano<28>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
            else if (st_ == 29) gelu_nano<29>(gsA, gk, uh[mt_][pr_], ul[mt_][pr_]);                                   \
        }
#define ACX_TOUCH2(h_, l_) asm volatile("" :: "v"(h_), "v"(l_));
#define ACX_BIAS_INIT(j_)                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * (j_) + 8 * q + 4 * hh);                 \
            _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                              \
                Xn[pt_][4 * q + 0] = bq[0]; Xn[pt_][4 * q + 1] = bq[1]; Xn[pt_][4 * q + 2] = bq[2]; Xn[pt_][4 * q + 3] = bq[3]; } \
        }
#define ACX_PACK_G()                                                                                            \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            gh[pt_][0] = __builtin_bit_cast(f32x4, uint4{uh[pt_][0], uh[pt_][1], uh[pt_][2], uh[pt_][3]});      \
            gh[pt_][1] = __builtin_bit_cast(f32x4, uint4{uh[pt_][4], uh[pt_][5], uh[pt_][6], uh[pt_][7]});      \
            gl[pt_][0] = __builtin_bit_cast(f32x4, uint4{ul[pt_][0], ul[pt_][1], ul[pt_][2], ul[pt_][3]});      \
            gl[pt_][1] = __builtin_bit_cast(f32x4, uint4{ul[pt_][4], ul[pt_][5], ul[pt_][6], ul[pt_][7]}); }
    // end of a segment: the pieces requested during it may stay in flight, everything older must have landed, and
    // every wave must be done reading the segment before its ring slot is requested again
#define ACX_SEG_END(issued_, stamp_)                                                                            \
        ACX_FENCE                                                                                               \
        ACX_WSTAMP(stamp_)                                                                                      \
        if (issued_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::kPieces) : "memory");                       \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                           \
        ACX_FENCE                                                                                               \
        ACX_WSTAMP(3)

    f32x16 Xn[PT], Xv[PT];        // Xn: pre-activation being accumulated by phase 1; Xv: the previous chunk's, input of the GELU
    f32x4 gh[PT][2], gl[PT][2];   // G(k - 1): B operand of phase 2, two k-steps, hi / lo halves
    unsigned uh[PT][8], ul[PT][8];
    GeluState2 gsA;

    // one phase-1 segment: X = b1 + W1c . LN(y)^T for chunk k_, image in ring slot grp_, requesting segment seg_ + 2
    auto phase1 = [&](auto with_gelu, const int k_, const int seg_, const int grp_) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;     // second half of the GELU of Xv rides on this segment's MFMAs
        const char* base = smem + grp_ * Cfg::kSegBytes;
        const bool dma = seg_ + 2 < Cfg::kSegs;
        const int g2 = (grp_ + 2) % 3;
        ACX_BIAS_INIT(k_)
        // fragments are read TWO units ahead of their MFMAs (three register sets rotating): with one unit of look-ahead
        // (96 matrix cycles) the LDS latency under load was exposed in front of every unit
        static_assert(Cfg::kSteps % 3 == 0, "the unit loop is unrolled by three");
        f32x4 f0h = ACX_W1_RD(base, 0, 0), f0l = ACX_W1_RD(base, 0, 1), f1h = ACX_W1_RD(base, 1, 0), f1l = ACX_W1_RD(base, 1, 1),
              f2h = ACX_W1_RD(base, 2, 0), f2l = ACX_W1_RD(base, 2, 1);
        // unit s_: MFMAs on the current set; in the gap behind its last MFMA the (counted) wait for the NEXT unit's set, then
        // the reads of unit s_ + 3 into the set just freed
#define ACX_P1_UNIT(s_, ch_, cl_, th_, tl_)                                                                     \
            ACX_FENCE                                                                                           \
            ACX_P1_MFMA(s_, ch_, cl_)                                                                           \
            if ((s_) + 1 < Cfg::kSteps) ACX_TOUCH2(th_, tl_)                                                    \
            if ((s_) + 3 < Cfg::kSteps) { ch_ = ACX_W1_RD(base, (s_) + 3, 0); cl_ = ACX_W1_RD(base, (s_) + 3, 1); } \
            ACX_FENCE
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; s += 3) {
            ACX_P1_UNIT(s, f0h, f0l, f1h, f1l)
            ACX_P1_UNIT(s + 1, f1h, f1l, f2h, f2l)
            ACX_P1_UNIT(s + 2, f2h, f2l, f0h, f0l)
        }
#undef ACX_P1_UNIT
        if constexpr (HV) { ACX_PACK_G() }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) Xv[pt] = Xn[pt];
        ACX_SEG_END(dma, 1)
    };
    // one phase-2 segment: out^T += W2c . G for the chunk whose G sits in gh / gl, image in ring slot grp_; with_gelu:
    // the first half of the GELU + split of Xv (the NEXT chunk) rides on this segment's MFMAs
    auto phase2 = [&](auto with_gelu, const int seg_, const int grp_) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;
        const char* base = smem + grp_ * Cfg::kSegBytes;
        const bool dma = seg_ + 2 < Cfg::kSegs;
        const int g2 = (grp_ + 2) % 3;
        static_assert(Cfg::kUnits % 3 == 0, "the unit loop is unrolled by three");
        f32x4 f0h = ACX_W2_RD(base, 0, 0), f0l = ACX_W2_RD(base, 0, 1), f1h = ACX_W2_RD(base, 1, 0), f1l = ACX_W2_RD(base, 1, 1),
              f2h = ACX_W2_RD(base, 2, 0), f2l = ACX_W2_RD(base, 2, 1);
#define ACX_P2_UNIT(i_, ch_, cl_, th_, tl_)                                                                     \
            ACX_FENCE                                                                                           \
            ACX_P2_MFMA(i_, ch_, cl_)                                                                           \
            if ((i_) + 1 < Cfg::kUnits) ACX_TOUCH2(th_, tl_)                                                    \
            if ((i_) + 3 < Cfg::kUnits) { ch_ = ACX_W2_RD(base, (i_) + 3, 0); cl_ = ACX_W2_RD(base, (i_) + 3, 1); } \
            ACX_FENCE
#pragma unroll
        for (int i = 0; i < Cfg::kUnits; i += 3) {
            ACX_P2_UNIT(i, f0h, f0l, f1h, f1l)
            ACX_P2_UNIT(i + 1, f1h, f1l, f2h, f2l)
            ACX_P2_UNIT(i + 2, f2h, f2l, f0h, f0l)
        }
#undef ACX_P2_UNIT
        ACX_SEG_END(dma, 2)
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // segments 0 and 1 landed ...
    __syncthreads();                                      // ... for every wave; b1s visible
    ACX_WSTAMP(0)                                         // tile prologue: y rows, LayerNorm, first two segments' flight
    // segment 0: phase 1 of chunk 0; the first half of its GELU has nothing to ride on
    phase1(std::false_type{}, 0, 0, 0);
    ACX_NANO_RANGE(0, 0, Cfg::kNano)
    // segments 2k-1 (phase 1 of chunk k + second half of GELU(k-1)) and 2k (phase 2 of chunk k-1 + first half of GELU(k));
    // ring slot = segment % 3
    int grp = 1;
    for (int k = 1; k < n - 1; ++k) {
        phase1(std::true_type{}, k, 2 * k - 1, grp);
        grp = grp == 2 ? 0 : grp + 1;
        phase2(std::true_type{}, 2 * k, grp);
        grp = grp == 2 ? 0 : grp + 1;
    }
    phase1(std::true_type{}, n - 1, 2 * n - 3, grp);
    grp = grp == 2 ? 0 : grp + 1;
    // the activations are dead from here on: their registers take the residual x of the tile, requested two segments
    // (~2 us) before the epilogue needs it
    float4 xr[PT][C / 8];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const float* xp = x + mrow[pt] * C + 4 * hh;
#pragma unroll
        for (int t = 0; t < C / 32; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[pt][4 * t + q] = *reinterpret_cast<const float4*>(xp + 32 * t + 8 * q);
    }
    phase2(std::true_type{}, 2 * n - 2, grp);
    grp = grp == 2 ? 0 : grp + 1;
    ACX_NANO_RANGE(1, 0, Cfg::kNano)       // second half of the last chunk's GELU: no phase-1 segment left to ride on
    ACX_PACK_G()
    phase2(std::false_type{}, 2 * n - 1, grp);
#undef ACX_WDMA
#undef ACX_H8
#undef ACX_FENCE
#undef ACX_W1_RD
#undef ACX_P1_MFMA
#undef ACX_W2_RD
#undef ACX_P2_MFMA
#undef ACX_TOUCH2
#undef ACX_BIAS_INIT
#undef ACX_NANO_RANGE
#undef ACX_AFTER_MFMA
#undef ACX_PACK_G
#undef ACX_SEG_END

    // ---- epilogue: lane (px, hh), tile t, q: channels 32t + 8q + 4hh .. +3  ->  x = x + out + b2 ---------
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
    if constexpr (LNOUT) {
        // last block of the stage in the full forward: the only reader of the new x is the LayerNorm in front of the
        // downsample conv (convnext.py:230-235): write its S16 operand instead (see mlp_fused_split.hip)
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < C / 32; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * t + 8 * q;
                const float4 bb = *reinterpret_cast<const float4*>(b2 + c + 4 * hh);
                const float4 v = xr[pt][4 * t + q];
                acc[pt][t][4 * q + 0] = v.x + fmaf(acc[pt][t][4 * q + 0], sinv2, bb.x);
                acc[pt][t][4 * q + 1] = v.y + fmaf(acc[pt][t][4 * q + 1], sinv2, bb.y);
                acc[pt][t][4 * q + 2] = v.z + fmaf(acc[pt][t][4 * q + 2], sinv2, bb.z);
                acc[pt][t][4 * q + 3] = v.w + fmaf(acc[pt][t][4 * q + 3], sinv2, bb.w);
                sum += (acc[pt][t][4 * q + 0] + acc[pt][t][4 * q + 1]) + (acc[pt][t][4 * q + 2] + acc[pt][t][4 * q + 3]);
            }
        }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int t = 0; t < C / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float u = acc[pt][t][r] - mean; d = fmaf(u, u, d); }
        d += __shfl_xor(d, 32);
        const float sc = kSplitLnScale / sqrtf(d * (1.0f / C) + 1e-6f);
        // lanes (px, 0) and (px, 1) trade halves so that each writes ONE 16-byte piece per block -- the lower lane the 8 hi
        // halves, the upper lane the 8 lo halves: 32 contiguous bytes per row and store instruction (the rows are cold: 8-byte
        // pieces cost a read-for-ownership of every sector)
        char* op = ln_out + mrow[pt] * (long long)(C * 4) + 16 * hh;
#pragma unroll
        for (int t = 0; t < C / 32; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned uhi[2], ulo[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    f32x2 v;
                    v.x = (acc[pt][t][4 * q + 2 * e] - mean) * sc; v.y = (acc[pt][t][4 * q + 2 * e + 1] - mean) * sc;
                    const h2 h = __builtin_convertvector(v, h2);
                    const f32x2 back = __builtin_convertvector(h, f32x2);
                    const h2 l = __builtin_convertvector(v - back, h2);
                    uhi[e] = __builtin_bit_cast(unsigned, h);
                    ulo[e] = __builtin_bit_cast(unsigned, l);
                    acx_pair_swap(uhi[e], ulo[e]);
                }
                // block of channels 32t + 8q .. +7: [8 hi][8 lo]
                if (valid[pt]) *reinterpret_cast<uint4*>(op + (4 * t + q) * 32) = uint4{uhi[0], uhi[1], ulo[0], ulo[1]};
            }
        }
    } else if (valid[pt]) {
        float* xp = x + mrow[pt] * C + 4 * hh;
#pragma unroll
        for (int t = 0; t < C / 32; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * t + 8 * q;
                const float4 bb = *reinterpret_cast<const float4*>(b2 + c + 4 * hh);
                float4 v = xr[pt][4 * t + q];
                v.x += fmaf(acc[pt][t][4 * q + 0], sinv2, bb.x);
                v.y += fmaf(acc[pt][t][4 * q + 1], sinv2, bb.y);
                v.z += fmaf(acc[pt][t][4 * q + 2], sinv2, bb.z);
                v.w += fmaf(acc[pt][t][4 * q + 3], sinv2, bb.w);
                *reinterpret_cast<float4*>(xp + c) = v;
            }
        }
    }
    }
    ACX_WSTAMP(4)
    ACX_WSTAMP_FLUSH
}

template <int C, int PT, bool LNOUT>
static int launch_wide_cfg(const BlockW& w, const float* y, float* x, long long M, void* ln_out, hipStream_t s) {
    using Cfg = WideCfg<C, PT>;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "weight ring does not fit the LDS");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_wide_kernel<C, PT, LNOUT>, kCuLdsBytes));
    const long long blocks = (M + Cfg::kPix - 1) / Cfg::kPix;
    mlp_fused_wide_kernel<C, PT, LNOUT><<<dim3((unsigned)blocks), dim3(Cfg::kThreads), kCuLdsBytes /* CU-exclusive */, s>>>(
        y, x, reinterpret_cast<const char*>(w.wstream_s), w.b1, w.b2, M, 1.0f / (kSplitLnScale * w.w1s_scale),
        1.0f / (w.hid_scale * w.w2s_scale), w.hid_scale, reinterpret_cast<char*>(ln_out));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

bool mlp_fused_wide_supported(int C) { return C == 384 || C == 192; }

int launch_mlp_fused_wide(acx_ctx* c, const BlockW& w, int C, const float* y, float* x, long long M, hipStream_t s,
                          void* ln_out) {
    if (!w.wstream_s) ACX_FAIL(ACX_ERR_STATE, "wide fused MLP: the weight stream was not packed for C=%d", C);
    ProfScope ps(c, ACX_K_MLP_WIDE, s);
    if (C == 384) return ln_out ? launch_wide_cfg<384, 1, true>(w, y, x, M, ln_out, s) : launch_wide_cfg<384, 1, false>(w, y, x, M, nullptr, s);
    if (C == 192) return ln_out ? launch_wide_cfg<192, 1, true>(w, y, x, M, ln_out, s) : launch_wide_cfg<192, 1, false>(w, y, x, M, nullptr, s);
    ACX_FAIL(ACX_ERR_SHAPE, "wide fused MLP: unsupported channel count %d", C);
}

}  // namespace acx
