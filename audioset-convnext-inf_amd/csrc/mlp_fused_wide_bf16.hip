// K4fb -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86) in the
// `bf16` arithmetic of BASELINE configs[2] ("bf16 with fp32 LayerNorm"): bf16 MFMA operands (LayerNorm output, weights,
// hidden activation), fp32 accumulation, fp32 LayerNorm statistics, GELU and residual stream.  The dataflow, the weight
// stream through a 3-slot LDS ring, the one-wave-per-SIMD schedule and the CU-exclusive launch are those of
// mlp_fused_wide.hip; what changes with ONE v_mfma_f32_32x32x16_bf16 per product instead of three fp16 ones:
//   * a chunk is 64 hidden units (two 32 x 32 pre-activation tiles): a segment -- the [64 x C] W1 image or the [C x 64]
//     W2 image of a chunk, 128 C bytes -- then carries as many MFMAs (C/8 per pixel tile) relative to its fixed costs
//     (LDS-DMA issues, barrier) as half a split segment does, instead of a third;
//   * the GELU is the 7 scalar micro-steps of split_math.h plus one bf16 pack (no lo half): 128 micro-steps per chunk and
//     pixel tile, half of them dealt over the MFMAs of each of the chunk's two segments.  At one MFMA per product the
//     vector work outweighs the matrix work for C <= 192: those stages are bound by the GELU's vector issue.
// Rounding points are those of the un-fused bf16 path (and of tests/test_gpu_bf16.py's emulation): LayerNorm output and
// GELU output rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32), weights rounded once at acx_finalize.
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int C, int PT>
struct WideBfCfg {
    static constexpr int kWaves = 4;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kPix = kWaves * 32 * PT;
    static constexpr int kChunks = 4 * C / 64;              // n: chunks of 64 hidden units
    static constexpr int kSegs = 2 * kChunks;
    static constexpr int kSegBytes = 128 * C;               // [64][C] or [C][64] bf16
    static constexpr int kPieces = kSegBytes / 1024 / kWaves;
    static constexpr int kSteps = C / 16;                   // k-steps of phase 1
    static constexpr int kTiles = C / 32;                   // out tiles of phase 2
    static constexpr int kUnits = 2 * kSteps;               // fragment reads of a segment: (k-step, X tile) or (out tile, k-step): 4 kTiles = 2 kSteps
    static constexpr int kMfmas = kUnits * PT;              // MFMAs per segment
    static constexpr int kHalf = 64 * PT;                   // GELU micro-steps a segment carries
    static constexpr size_t kLdsBytes = 3 * (size_t)kSegBytes + 4 * C * 4 + C * 4;      // ring, b1, b2
    static_assert(C % 32 == 0 && kUnits % kPieces == 0, "unit / piece bookkeeping");
    // W1 rows are 2 C bytes = C/8 chunks of 16 B: the XOR that spreads 16 consecutive rows over the LDS banks
    static constexpr int kSwzBits = (C % 128 == 0) ? 4 : ((C % 64 == 0) ? 3 : 2);
    __host__ __device__ static int swz1(int row) { return kSwzBits == 4 ? (row & 15) : (kSwzBits == 3 ? ((row >> 1) & 7) : ((row >> 2) & 3)); }
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    f32x2 v; v.x = a; v.y = b;
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, h);
}

// ABF: y and x are bf16 in HBM (ACX_PREC_BF16_ACT): 8 channels per 16-byte load, 4 per 8-byte load / store; everything between
// the loads and the stores is unchanged (fp32 LayerNorm statistics, fp32 accumulate, fp32 GELU, fp32 residual add)
template <bool ABF>
__device__ __forceinline__ void acx_ld8(const void* base, long long off, float* o) {        // 8 consecutive channels
    if constexpr (ABF) {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + off);
        o[0] = acx_bf16_lo(u.x); o[1] = acx_bf16_hi(u.x); o[2] = acx_bf16_lo(u.y); o[3] = acx_bf16_hi(u.y);
        o[4] = acx_bf16_lo(u.z); o[5] = acx_bf16_hi(u.z); o[6] = acx_bf16_lo(u.w); o[7] = acx_bf16_hi(u.w);
    } else {
        const float4 v0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        const float4 v1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
        o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
    }
}
template <bool ABF>
__device__ __forceinline__ float4 acx_ld4(const void* base, long long off) {                 // 4 consecutive channels
    if constexpr (ABF) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(base) + off);
        return make_float4(acx_bf16_lo(u.x), acx_bf16_hi(u.x), acx_bf16_lo(u.y), acx_bf16_hi(u.y));
    } else {
        return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    }
}
template <bool ABF>
__device__ __forceinline__ void acx_st4(void* base, long long off, float4 v) {
    if constexpr (ABF) *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(base) + off) = uint2{acx_pack_bf16x2(v.x, v.y), acx_pack_bf16x2(v.z, v.w)};
    else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = v;
}

template <int C, int PT, bool LNOUT, bool ABF>
__global__ __launch_bounds__(256) void mlp_fused_wide_bf16_kernel(
    const void* __restrict__ y, void* __restrict__ x, const char* __restrict__ wstream /*[2n][128 C bytes]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, int ld_out,
    __bf16* __restrict__ ln_out /* LNOUT: (M, ld_out) bf16 rows of LayerNorm(x_new), written INSTEAD of x */) {
    using Cfg = WideBfCfg<C, PT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + 3 * Cfg::kSegBytes);   // [4C]
    float* b2s = b1s + 4 * C;     // [C]: read by the epilogue (round 4: as loads from b2 inside its store loop they were one L2
    // round trip per store -- hipcc waits for every outstanding vector-memory operation there, the previous store included)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    ACX_CLAIM_VGPR(255);          // CU-exclusive: one wave per SIMD holds the SIMD's whole register file
    ACX_CLAIM_AGPR(255);
    long long mrow[PT];
    bool valid[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        mrow[pt] = (long long)blockIdx.x * Cfg::kPix + (wave * PT + pt) * 32 + l31;
        valid[pt] = mrow[pt] < M;
        if (!valid[pt]) mrow[pt] = M - 1;
    }

    constexpr int n = Cfg::kChunks;
    // LDS-DMA pieces with a SCALAR base and a 32-bit lane offset (split_math.h acx_glds16_s, round 5): a piece's source is
    // contiguous, so the 64-bit address per lane of the builtin form is not needed -- and that form's issue was 15 % of the paired
    // kernel (profiles/r05_e_lds_dma_saddr.txt).  Issued from inline asm: the counted waits at the segment ends are this file's own.
    const unsigned smem_a = acx_lds_addr(smem);
    const unsigned dma_voff = lane * 16;
#define ACX_WDMA(seg_, piece_, grp_)                                                                             \
        acx_glds16_s(wstream + (long long)(seg_) * Cfg::kSegBytes + (wave * Cfg::kPieces + (piece_)) * 1024, dma_voff,   \
                     smem_a + (unsigned)((grp_) * Cfg::kSegBytes + (wave * Cfg::kPieces + (piece_)) * 1024));
    // In the segment loops the wave's kPieces pieces go out as RUNS (split_math.h acx_glds16_run_s): they are 1 KB apart in the stream
    // and in the LDS, so one M0 write and one scalar base serve up to eight of them and a piece is ONE instruction instead of six
    // (round 6: 144 -> 32 instructions per chunk at C = 384 in a loop that is bound by its instruction count).
    // Nothing else touches M0 between the pieces of a run.
#define ACX_WDMA_RUN(seg_, p_, grp_) {                                                                           \
        const int run0_ = ((p_) / 8) * 8;                                                                       \
        if ((p_) == run0_) acx_set_m0(smem_a + (unsigned)((grp_) * Cfg::kSegBytes + (wave * Cfg::kPieces + run0_ + 4) * 1024)); \
        acx_glds16_run_s(wstream + (long long)(seg_) * Cfg::kSegBytes + (wave * Cfg::kPieces + run0_ + 4) * 1024, dma_voff, (p_) - run0_); }
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(0, p, 0)
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(1, p, 1)
    // bias staging: 16 bytes per lane, requested here (ahead of the tile's rows: vmcnt completes in order) and written to the LDS
    // behind the LayerNorm, so that it has no wait of its own
    constexpr int kB1Iters = (C + Cfg::kThreads - 1) / Cfg::kThreads;
    float4 b1v[kB1Iters];
#pragma unroll
    for (int q = 0; q < kB1Iters; ++q) {
        const int i = q * Cfg::kThreads + tid;
        b1v[q] = reinterpret_cast<const float4*>(b1)[i < C ? i : 0];
    }
    static_assert(C / 4 <= Cfg::kThreads, "one float4 of b2 per thread");
    const float4 b2v = reinterpret_cast<const float4*>(b2)[tid < C / 4 ? tid : 0];

    // ---- this wave's activations: lane (px = l31, half hh) holds channels 16s + 8hh .. +7 as 8 bf16, s = 0..C/16-1 ----
    f32x4 act[PT][Cfg::kSteps];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        float a[C / 2];
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) acx_ld8<ABF>(y, mrow[pt] * C + 8 * hh + 16 * s, a + 8 * s);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < C / 2; ++i) sum += a[i];
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < C / 2; ++i) { const float t = a[i] - mean; d = fmaf(t, t, d); }
        d += __shfl_xor(d, 32);
        const float rstd = 1.0f / sqrtf(d * (1.0f / C) + 1e-6f);
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) {
            unsigned u4[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) u4[p] = pack_bf16((a[8 * s + 2 * p] - mean) * rstd, (a[8 * s + 2 * p + 1] - mean) * rstd);
            act[pt][s] = __builtin_bit_cast(f32x4, uint4{u4[0], u4[1], u4[2], u4[3]});
        }
    }

#pragma unroll
    for (int q = 0; q < kB1Iters; ++q) {
        const int i = q * Cfg::kThreads + tid;
        if (i < C) reinterpret_cast<float4*>(b1s)[i] = float4{0.5f * b1v[q].x, 0.5f * b1v[q].y, 0.5f * b1v[q].z, 0.5f * b1v[q].w};   // z = 0.5 v (gelu2h_micro)
    }
    if (tid < C / 4) reinterpret_cast<float4*>(b2s)[tid] = b2v;

    f32x16 acc[PT][Cfg::kTiles];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pt][t][r] = 0.f;

    // fragment addresses inside a segment (without the ring offset):
    //   W1 image: row = hidden unit of the chunk (0..63), 2 C bytes = C/8 chunks of 16 B; X tile j uses rows 32 j + l31; the
    //             fragment of k-step s is chunk p = 2 s + hh at position p ^ swz1(row) (swz1(32 j + l31) == swz1(l31))
    //   W2 image: row = out channel (128 B = 8 chunks), tile t rows 32 t + l31, chunk 2 s' + hh at position ^ ((l31 >> 1) & 7)
    constexpr int kVar1 = 1 << (Cfg::kSwzBits - 1);          // k-steps whose chunk index differs in the XORed low bits
    int w1off[kVar1], w2off[4];
#pragma unroll
    for (int q = 0; q < kVar1; ++q) w1off[q] = l31 * (2 * C) + (((2 * q + hh) ^ Cfg::swz1(l31)) << 4);
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) w2off[sp] = l31 * 128 + (((2 * sp + hh) ^ ((l31 >> 1) & 7)) << 4);
    const GeluK3 gk = gelu_k2h();      // X holds z = 0.5 v itself: 0.5 W1 in the stream, 0.5 b1 staged (split_math.h, gelu2h_micro)

#define ACX_B8(v_) __builtin_bit_cast(bf16x8, v_)
#define ACX_FENCE __builtin_amdgcn_sched_barrier(0);
    // phase-1 unit u = (k-step s = u >> 1, X tile j = u & 1)
#define ACX_W1_RD(base_, u_) (*reinterpret_cast<const f32x4*>((base_) + ((u_) & 1) * (32 * 2 * C) + (((u_) >> 1) / kVar1) * (kVar1 * 32) + w1off[((u_) >> 1) % kVar1]))
    // MFMA number m_ of a segment is followed (behind a scheduling fence) by its share of the kHalf GELU micro-steps
#define ACX_AFTER_MFMA(HV_, half_, m_, X_)                                                                      \
        ACX_FENCE if constexpr (HV_) { ACX_MICRO_RANGE(half_, Cfg::kHalf * (m_) / Cfg::kMfmas, Cfg::kHalf * ((m_) + 1) / Cfg::kMfmas, X_) } ACX_FENCE
#define ACX_P1_MFMA(u_, f_)                                                                                     \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            Xb[CUR][pt_][(u_) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ACX_B8(f_), ACX_B8(act[pt_][(u_) >> 1]), Xb[CUR][pt_][(u_) & 1], 0, 0, 0); \
            ACX_AFTER_MFMA(HV, 1, (u_) * PT + pt_, Xb[1 - CUR]) }
    // phase-2 unit u = (out tile t = u >> 2, k-step s' = u & 3)
#define ACX_W2_RD(base_, u_) (*reinterpret_cast<const f32x4*>((base_) + ((u_) >> 2) * 4096 + w2off[(u_) & 3]))
#define ACX_P2_MFMA(u_, f_)                                                                                     \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                                  \
            acc[pt_][(u_) >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ACX_B8(f_), ACX_B8(g[pt_][(u_) & 3]), acc[pt_][(u_) >> 2], 0, 0, 0); \
            ACX_AFTER_MFMA(HV, 0, (u_) * PT + pt_, Xb[CUR]) }
    // micro-steps [from, to) of the kHalf that segment half half_ carries: step sg_ belongs to pixel tile sg_ / 64; half h
    // is X tile j = h of the chunk: 8 register pairs x 8 steps (7 of the GELU, 1 bf16 pack) turn Xv[.][j] into un[.][j]
#define ACX_MICRO_RANGE(half_, from_, to_, X_)                                                                  \
        _Pragma("unroll") for (int sg_ = (from_); sg_ < (to_); ++sg_) {                                         \
            const int mt_ = sg_ / 64, pr_ = (sg_ % 64) / 8, st_ = sg_ % 8;                                      \
            const float ax_ = X_[mt_][half_][2 * pr_], ay_ = X_[mt_][half_][2 * pr_ + 1];                       \
            if (st_ == 0) gelu2h_micro<0>(gs, gk, ax_, ay_);                                                     \
            else if (st_ == 1) gelu2h_micro<1>(gs, gk, ax_, ay_);                                                \
            else if (st_ == 2) gelu2h_micro<2>(gs, gk, ax_, ay_);                                                \
            else if (st_ == 3) gelu2h_micro<3>(gs, gk, ax_, ay_);                                                \
            else if (st_ == 4) gelu2h_micro<4>(gs, gk, ax_, ay_);                                                \
            else if (st_ == 5) gelu2h_micro<5>(gs, gk, ax_, ay_);                                                \
            else if (st_ == 6) gelu2h_micro<6>(gs, gk, ax_, ay_);                                                \
            else un[mt_][half_][pr_] = pack_bf16(gs.qx, gs.qy);                                                 \
        }
#define ACX_TOUCH1(f_) { asm volatile("" :: "v"(f_)); }
    // Xn[.][j] starts from the bias of hidden units 64 k + 32 j + (lane layout of a 32 x 32 accumulator)
#define ACX_BIAS_INIT(k_)                                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 64 * (k_) + 32 * j_ + 8 * q + 4 * hh);       \
            _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_) {                                              \
                Xb[CUR][pt_][j_][4 * q + 0] = bq[0]; Xb[CUR][pt_][j_][4 * q + 1] = bq[1]; Xb[CUR][pt_][j_][4 * q + 2] = bq[2]; Xb[CUR][pt_][j_][4 * q + 3] = bq[3]; } \
        }
    // G of the chunk: k-step s' = 2 j + (pair >> 2) of phase 2 takes pairs 4 (s' & 1) .. + 3 of X tile j
#define ACX_PACK_G()                                                                                            \
        _Pragma("unroll") for (int pt_ = 0; pt_ < PT; ++pt_)                                                    \
        _Pragma("unroll") for (int sp_ = 0; sp_ < 4; ++sp_)                                                     \
            g[pt_][sp_] = __builtin_bit_cast(f32x4, uint4{un[pt_][sp_ >> 1][4 * (sp_ & 1) + 0], un[pt_][sp_ >> 1][4 * (sp_ & 1) + 1], \
                                                          un[pt_][sp_ >> 1][4 * (sp_ & 1) + 2], un[pt_][sp_ >> 1][4 * (sp_ & 1) + 3]});
#define ACX_SEG_END(issued_)                                                                                    \
        ACX_FENCE                                                                                               \
        if (issued_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::kPieces) : "memory");                       \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                           \
        ACX_FENCE

    // Two sets of pre-activation tiles in turn (round 6): phase 1 of chunk k accumulates into Xb[k & 1] while the second half of the
    // GELU of chunk k - 1 reads Xb[(k - 1) & 1]; phase 2 then reads the first half of Xb[k & 1].  (Until round 5 one set was the
    // accumulator and was copied to the other at the end of every phase 1: 32 v_accvgpr_read + 32 v_accvgpr_write per chunk and
    // pixel tile, 8 % of the loop's instructions, in a loop bound by their count -- DESIGN.md 3i.)
    f32x16 Xb[2][PT][2];
    f32x4 g[PT][4];               // G(k - 1): B operand of phase 2, four k-steps of 8 bf16
    unsigned un[PT][2][8];        // G(k) under construction
    constexpr int kDmaStride = Cfg::kUnits / Cfg::kPieces;
    GeluState3 gs;

    auto phase1 = [&](auto with_gelu, auto parity, const int k_, const int seg_, const int grp_) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;     // second half (X tile 1) of the GELU of the chunk before rides on this segment's MFMAs
        constexpr int CUR = decltype(parity)::value;        // k_ & 1
        const char* base = smem + grp_ * Cfg::kSegBytes;
        const bool dma = seg_ + 2 < Cfg::kSegs;
        const int g2 = (grp_ + 2) % 3;
        ACX_BIAS_INIT(k_)
        // fragment reads run TWO units ahead of their MFMA (three rotating buffers): at one MFMA per fragment the ~100
        // cycles of a ds_read_b128 do not fit behind a single 32-cycle MFMA and its share of the GELU
        f32x4 f[3];
        f[0] = ACX_W1_RD(base, 0);
        f[1] = ACX_W1_RD(base, 1);
#pragma unroll
        for (int u = 0; u < Cfg::kUnits; ++u) {
            if (u + 2 < Cfg::kUnits) f[(u + 2) % 3] = ACX_W1_RD(base, u + 2);
            ACX_FENCE
            ACX_P1_MFMA(u, f[u % 3])
            if (u % kDmaStride == 0 && dma) { ACX_WDMA_RUN(seg_ + 2, u / kDmaStride, g2) }
            ACX_FENCE
            if (u + 1 < Cfg::kUnits) ACX_TOUCH1(f[(u + 1) % 3])
        }
        if constexpr (HV) { ACX_PACK_G() }
        ACX_SEG_END(dma)
    };
    auto phase2 = [&](auto with_gelu, auto parity, const int seg_, const int grp_) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;     // first half (X tile 0) of the GELU of Xb[CUR] (the NEXT chunk) rides here
        constexpr int CUR = decltype(parity)::value;
        const char* base = smem + grp_ * Cfg::kSegBytes;
        const bool dma = seg_ + 2 < Cfg::kSegs;
        const int g2 = (grp_ + 2) % 3;
        f32x4 f[3];
        f[0] = ACX_W2_RD(base, 0);
        f[1] = ACX_W2_RD(base, 1);
#pragma unroll
        for (int u = 0; u < Cfg::kUnits; ++u) {
            if (u + 2 < Cfg::kUnits) f[(u + 2) % 3] = ACX_W2_RD(base, u + 2);
            ACX_FENCE
            ACX_P2_MFMA(u, f[u % 3])
            if (u % kDmaStride == 0 && dma) { ACX_WDMA_RUN(seg_ + 2, u / kDmaStride, g2) }
            ACX_FENCE
            if (u + 1 < Cfg::kUnits) ACX_TOUCH1(f[(u + 1) % 3])
        }
        ACX_SEG_END(dma)
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // segments 0 and 1 landed (the pieces are issued from asm: hipcc does not wait for them)
    __syncthreads();      // ... in every wave; b1s visible
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    static_assert(n % 2 == 0, "the chunk loop runs two chunks per iteration (the two X sets in turn)");
    phase1(std::false_type{}, P0{}, 0, 0, 0);
    ACX_MICRO_RANGE(0, 0, Cfg::kHalf, Xb[0])
    int grp = 1;
    for (int k = 1; k < n - 1; k += 2) {
        phase1(std::true_type{}, P1{}, k, 2 * k - 1, grp);
        grp = grp == 2 ? 0 : grp + 1;
        phase2(std::true_type{}, P1{}, 2 * k, grp);
        grp = grp == 2 ? 0 : grp + 1;
        phase1(std::true_type{}, P0{}, k + 1, 2 * k + 1, grp);
        grp = grp == 2 ? 0 : grp + 1;
        phase2(std::true_type{}, P0{}, 2 * k + 2, grp);
        grp = grp == 2 ? 0 : grp + 1;
    }
    phase1(std::true_type{}, P1{}, n - 1, 2 * n - 3, grp);
    grp = grp == 2 ? 0 : grp + 1;
    // the activations are dead from here on: their registers take the residual x of the tile
    // (not for LNOUT at C = 384: its epilogue needs every accumulator AND every residual value at once for the LayerNorm
    // statistics -- 288 + working registers: hipcc spilled 640 B per lane around the last segments; the residual of that
    // variant is read in the epilogue instead, tile by tile, from rows the previous block left in the cache)
    // kXM: the residual through the matrix pipe (below) -- where it measured faster: C = 192 (two pixel tiles per wave: 205 -> 198 us
    // per block in the lab, profiles/r05_c_ring_residual_ab.txt; C = 384: 170 -> 174, hipcc spills fragments that are in flight
    // during the last segments; C = 96: no change)
    constexpr bool kXM = ABF && C == 192;
    constexpr bool kPrefetchX = !(LNOUT && C >= 384) || kXM;
    // bf16 rows (ABF, round 5): the residual enters through the matrix pipe -- out += I . x with x read as B fragments (lane
    // (px, hh): channels 16 u + 8 hh .. + 7, ONE 16-byte load) and I the identity as an A fragment: 1.0 x bf16 is exact in the fp32
    // accumulate, two MFMAs per out tile.  In the accumulator layout the rows were 48 eight-byte loads per pixel tile, and a
    // load's address processing costs 4 cycles per LANE whatever its width (tools/lab/pair_lab.hip stamps: 47 k cycles of the
    // CU's one texture-address path per 128-pixel tile, the path the LDS-DMA weight pieces queue on) -- now 24 sixteen-byte
    // loads into half the registers (C/4 per pixel tile instead of C/2).
    float4 xr[PT][(kPrefetchX && !kXM) ? C / 8 : 1];
    f32x4 xf[PT][kXM ? C / 16 : 1];
    constexpr int kEarlyX = kXM ? C / 16 : 0;
    f32x4 ident[2];
    if constexpr (kXM) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const int e = l31 - 16 * sp - 8 * hh;             // this lane's row of I has its 1 at k = 16 s' + 8 hh + e
            unsigned w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) w[p] = (e == 2 * p ? 0x3f80u : 0u) | (e == 2 * p + 1 ? 0x3f800000u : 0u);
            ident[sp] = __builtin_bit_cast(f32x4, uint4{w[0], w[1], w[2], w[3]});
        }
    }
    // C = 384: requested one segment later, once the pre-activation tiles of the last GELU are dead (two segments ahead
    // hipcc spilled 156 B per lane around here)
    constexpr bool kLateX = C >= 384;
#define ACX_LOAD_XR()                                                                                           \
    if constexpr (kXM) {                                                                                        \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt)                                                       \
        _Pragma("unroll") for (int u = 0; u < kEarlyX; ++u)                                                     \
            xf[pt][u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(x) + mrow[pt] * C + 8 * hh + 16 * u); \
    } else if constexpr (kPrefetchX) {                                                                          \
        _Pragma("unroll") for (int pt = 0; pt < PT; ++pt)                                                       \
        _Pragma("unroll") for (int t = 0; t < Cfg::kTiles; ++t)                                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) xr[pt][4 * t + q] = acx_ld4<ABF>(x, mrow[pt] * C + 4 * hh + 32 * t + 8 * q); \
    }
    if constexpr (!kLateX) { ACX_LOAD_XR() }
    phase2(std::true_type{}, P1{}, 2 * n - 2, grp);
    grp = grp == 2 ? 0 : grp + 1;
    ACX_MICRO_RANGE(1, 0, Cfg::kHalf, Xb[1])       // second half of the last chunk's GELU: no phase-1 segment left to ride on
    ACX_PACK_G()
    if constexpr (kLateX) { ACX_LOAD_XR() }
    phase2(std::false_type{}, P1{}, 2 * n - 1, grp);
#undef ACX_LOAD_XR
#undef ACX_WDMA
#undef ACX_WDMA_RUN
#undef ACX_B8
#undef ACX_FENCE
#undef ACX_W1_RD
#undef ACX_AFTER_MFMA
#undef ACX_P1_MFMA
#undef ACX_W2_RD
#undef ACX_P2_MFMA
#undef ACX_MICRO_RANGE
#undef ACX_TOUCH1
#undef ACX_BIAS_INIT
#undef ACX_PACK_G
#undef ACX_SEG_END

    // ---- epilogue: lane (px, hh), tile t, q: channels 32t + 8q + 4hh .. +3  ->  x = x + out + b2 ---------
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        if constexpr (kXM) {        // out += I . x
#pragma unroll
            for (int u = kEarlyX; u < C / 16; ++u)
                xf[pt][u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(x) + mrow[pt] * C + 8 * hh + 16 * u);
#pragma unroll
            for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
                    acc[pt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ident[sp]), __builtin_bit_cast(bf16x8, xf[pt][2 * t + sp]), acc[pt][t], 0, 0, 0);
        }
        if constexpr (LNOUT) {
            // last block of the stage in the full forward: the only reader of the new x is the LayerNorm in front of the
            // downsample conv (convnext.py:230-235): write that GEMM's bf16 operand rows instead of x
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bb = *reinterpret_cast<const float4*>(b2s + 32 * t + 8 * q + 4 * hh);
                    float4 v;
                    if constexpr (kXM) v = make_float4(0.f, 0.f, 0.f, 0.f);         // (already in acc)
                    else if constexpr (kPrefetchX) v = xr[pt][4 * t + q];
                    else v = acx_ld4<ABF>(x, mrow[pt] * C + 4 * hh + 32 * t + 8 * q);
                    acc[pt][t][4 * q + 0] += v.x + bb.x; acc[pt][t][4 * q + 1] += v.y + bb.y;
                    acc[pt][t][4 * q + 2] += v.z + bb.z; acc[pt][t][4 * q + 3] += v.w + bb.w;
                    sum += (acc[pt][t][4 * q + 0] + acc[pt][t][4 * q + 1]) + (acc[pt][t][4 * q + 2] + acc[pt][t][4 * q + 3]);
                }
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / C);
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float u = acc[pt][t][r] - mean; d = fmaf(u, u, d); }
            d += __shfl_xor(d, 32);
            const float rstd = 1.0f / sqrtf(d * (1.0f / C) + 1e-6f);
            {
                // lanes (px, 0) and (px, 1) trade pieces so that each writes 16 bytes (8 channels) per store: the lower lane the
                // group q = 2j, the upper lane the group q = 2j + 1 -- 32 contiguous bytes per row and store instruction (the
                // rows are cold: 8-byte pieces cost a read-for-ownership of every sector)
                __bf16* op = ln_out + mrow[pt] * (long long)ld_out + 8 * hh;
#pragma unroll
                for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        unsigned e[2], o[2];        // this lane's 4 channels of group 2j / 2j + 1
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            e[w] = pack_bf16((acc[pt][t][8 * j + 2 * w] - mean) * rstd, (acc[pt][t][8 * j + 2 * w + 1] - mean) * rstd);
                            o[w] = pack_bf16((acc[pt][t][8 * j + 4 + 2 * w] - mean) * rstd, (acc[pt][t][8 * j + 4 + 2 * w + 1] - mean) * rstd);
                            acx_pair_swap(e[w], o[w]);
                        }
                        if (valid[pt]) *reinterpret_cast<uint4*>(op + 32 * t + 16 * j) = uint4{e[0], e[1], o[0], o[1]};
                    }
                // the row is ld_out = pad64(C) elements long (the GEMM's K is padded with zero weights): zero the padding, or
                // stale bytes that happen to be NaN would poison the products
                if (valid[pt])
                    for (int c = C + 8 * hh; c < ld_out; c += 16)
                        *reinterpret_cast<uint4*>(ln_out + mrow[pt] * (long long)ld_out + c) = uint4{0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float4 v[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int q = 2 * j + i;
                        const float4 bb = *reinterpret_cast<const float4*>(b2s + 32 * t + 8 * q + 4 * hh);
                        if constexpr (kXM) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);          // (already in acc)
                        else v[i] = xr[pt][4 * t + q];
                        v[i].x += acc[pt][t][4 * q + 0] + bb.x;
                        v[i].y += acc[pt][t][4 * q + 1] + bb.y;
                        v[i].z += acc[pt][t][4 * q + 2] + bb.z;
                        v[i].w += acc[pt][t][4 * q + 3] + bb.w;
                    }
                    if constexpr (ABF) {        // bf16 rows: 16-byte pieces by trading halves with the partner lane (as in LNOUT)
                        unsigned e[2] = {acx_pack_bf16x2(v[0].x, v[0].y), acx_pack_bf16x2(v[0].z, v[0].w)};
                        unsigned o[2] = {acx_pack_bf16x2(v[1].x, v[1].y), acx_pack_bf16x2(v[1].z, v[1].w)};
                        acx_pair_swap(e[0], o[0]);
                        acx_pair_swap(e[1], o[1]);
                        if (valid[pt])
                            *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(x) + mrow[pt] * C + 8 * hh + 32 * t + 16 * j) = uint4{e[0], e[1], o[0], o[1]};
                    } else if (valid[pt]) {
                        acx_st4<false>(x, mrow[pt] * C + 4 * hh + 32 * t + 16 * j, v[0]);
                        acx_st4<false>(x, mrow[pt] * C + 4 * hh + 32 * t + 16 * j + 8, v[1]);
                    }
                }
        }
    }
}

template <int C, int PT, bool LNOUT, bool ABF>
static int launch_wide_bf16_cfg(const BlockW& w, const void* y, void* x, long long M, void* ln_out, int ld_out, hipStream_t s) {
    using Cfg = WideBfCfg<C, PT>;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "weight ring does not fit the LDS");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_wide_bf16_kernel<C, PT, LNOUT, ABF>, kCuLdsBytes));
    const long long blocks = (M + Cfg::kPix - 1) / Cfg::kPix;
    launch_kernel(&mlp_fused_wide_bf16_kernel<C, PT, LNOUT, ABF>, dim3((unsigned)blocks), dim3(Cfg::kThreads), kCuLdsBytes /* CU-exclusive */, s,
        y, x, reinterpret_cast<const char*>(w.wstream_b), w.b1, w.b2, M, ld_out, reinterpret_cast<__bf16*>(ln_out));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// ---- C = 96: weights-stationary variant -------------------------------------------------------------------------------
// At C = 96 the WHOLE bf16 weight stream (12 segments x 12 KB = 144 KB) fits the LDS.  A persistent workgroup (one per
// CU, 8 waves) loads it once; after that its waves never synchronise again: each walks its own 32-pixel tiles
// (LayerNorm -> 6 chunks of [phase 1, GELU, phase 2] -> residual store) at its own pace.  What that buys over the ring
// kernel above: its workgroups all moved through "load the tile / compute / store the tile" in step, so the HBM phase
// (3 tensor passes, ~190 us per block at the HBM rate) and the compute phase (~210 us) added up; free-running waves
// spread over all phases, memory latency hides behind the other waves' matrix and vector work, and the per-segment
// barrier and LDS-DMA issue are gone.  Same stream layout (api.hip packs one `wstream_b` for both kernels), same
// arithmetic and rounding points.
template <bool LNOUT, bool ABF>
__global__ __launch_bounds__(512) void mlp_fused_stat_bf16_kernel(
    const void* __restrict__ y, void* __restrict__ x, const char* __restrict__ wstream /*[12][12288 B]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, int ld_out, __bf16* __restrict__ ln_out) {
    constexpr int C = 96;
    using Cfg = WideBfCfg<C, 1>;
    constexpr int n = Cfg::kChunks;                           // 6 chunks of 64 hidden units
    constexpr int kStreamBytes = Cfg::kSegs * Cfg::kSegBytes; // 147 456
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + kStreamBytes);   // [4C]
    float* b2s = b1s + 4 * C;                                     // [C]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    ACX_CLAIM_VGPR(255);          // CU-exclusive: 8 waves x 256 registers hold the SIMDs' whole register files

    // the stream into the LDS: 144 pieces of 1 KB, 18 per wave
#pragma unroll
    for (int p = 0; p < kStreamBytes / 1024 / 8; ++p) {
        const int piece = wave * (kStreamBytes / 1024 / 8) + p;
        acx_glds16_s(wstream + piece * 1024, lane * 16, acx_lds_addr(smem) + (unsigned)(piece * 1024));
    }
    for (int i = tid; i < 4 * C; i += 512) b1s[i] = 0.5f * b1[i];        // z = 0.5 v (gelu2h_micro)
    if (tid < C) b2s[tid] = b2[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr int kVar1 = 1 << (Cfg::kSwzBits - 1);
    int w1off[kVar1], w2off[4];
#pragma unroll
    for (int q = 0; q < kVar1; ++q) w1off[q] = l31 * (2 * C) + (((2 * q + hh) ^ Cfg::swz1(l31)) << 4);
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) w2off[sp] = l31 * 128 + (((2 * sp + hh) ^ ((l31 >> 1) & 7)) << 4);
    const GeluK3 gk = gelu_k2h();      // X holds z = 0.5 v itself: 0.5 W1 in the stream, 0.5 b1 staged (split_math.h, gelu2h_micro)
#define ACX_B8(v_) __builtin_bit_cast(bf16x8, v_)
#define ACX_W1_RD(base_, u_) (*reinterpret_cast<const f32x4*>((base_) + ((u_) & 1) * (32 * 2 * C) + (((u_) >> 1) / kVar1) * (kVar1 * 32) + w1off[((u_) >> 1) % kVar1]))
#define ACX_W2_RD(base_, u_) (*reinterpret_cast<const f32x4*>((base_) + ((u_) >> 2) * 4096 + w2off[(u_) & 3]))

    const long long ntiles = (M + 31) / 32;
    const long long tstride = (long long)gridDim.x * 8;
    // software pipeline over the wave's tiles: the y rows of tile i + 1 are requested before the chunks of tile i start
    // (raw 16-byte loads: fp32 two per k-step, bf16 one)
    float4 yn[ABF ? C / 16 : C / 8];
#define ACX_LOAD_Y(tile_)                                                                                       \
    {   long long r_ = (tile_) * 32 + l31; if (r_ >= M) r_ = M - 1;                                             \
        if constexpr (ABF) {                                                                                    \
            const __bf16* yp_ = reinterpret_cast<const __bf16*>(y) + r_ * C + 8 * hh;                           \
            _Pragma("unroll") for (int s = 0; s < Cfg::kSteps; ++s) yn[s] = *reinterpret_cast<const float4*>(yp_ + 16 * s); \
        } else {                                                                                                \
            const float* yp_ = reinterpret_cast<const float*>(y) + r_ * C + 8 * hh;                             \
            _Pragma("unroll") for (int s = 0; s < Cfg::kSteps; ++s) {                                           \
                yn[2 * s] = *reinterpret_cast<const float4*>(yp_ + 16 * s); yn[2 * s + 1] = *reinterpret_cast<const float4*>(yp_ + 16 * s + 4); } } }
    long long tile = (long long)blockIdx.x * 8 + wave;
    if (tile < ntiles) ACX_LOAD_Y(tile)
    for (; tile < ntiles; tile += tstride) {
        long long mrow = tile * 32 + l31;
        const bool valid = mrow < M;
        if (!valid) mrow = M - 1;
        // ---- LayerNorm of the tile's rows: lane (px = l31, half hh) holds channels 16 s + 8 hh .. + 7 ----
        f32x4 act[Cfg::kSteps];
        {
            float a[C / 2];
            if constexpr (ABF) {
#pragma unroll
                for (int i = 0; i < C / 16; ++i) {
                    const uint4 u = __builtin_bit_cast(uint4, yn[i]);
                    a[8 * i + 0] = acx_bf16_lo(u.x); a[8 * i + 1] = acx_bf16_hi(u.x); a[8 * i + 2] = acx_bf16_lo(u.y); a[8 * i + 3] = acx_bf16_hi(u.y);
                    a[8 * i + 4] = acx_bf16_lo(u.z); a[8 * i + 5] = acx_bf16_hi(u.z); a[8 * i + 6] = acx_bf16_lo(u.w); a[8 * i + 7] = acx_bf16_hi(u.w);
                }
            } else {
#pragma unroll
                for (int i = 0; i < C / 8; ++i) { a[4 * i] = yn[i].x; a[4 * i + 1] = yn[i].y; a[4 * i + 2] = yn[i].z; a[4 * i + 3] = yn[i].w; }
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
            const float rstd = 1.0f / sqrtf(d * (1.0f / C) + 1e-6f);
#pragma unroll
            for (int s = 0; s < Cfg::kSteps; ++s) {
                unsigned u4[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) u4[p] = pack_bf16((a[8 * s + 2 * p] - mean) * rstd, (a[8 * s + 2 * p + 1] - mean) * rstd);
                act[s] = __builtin_bit_cast(f32x4, uint4{u4[0], u4[1], u4[2], u4[3]});
            }
        }
        // the residual rows: in flight while the chunks run
        float4 xr[C / 8];
        {
#pragma unroll
            for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) xr[4 * t + q] = acx_ld4<ABF>(x, mrow * C + 4 * hh + 32 * t + 8 * q);
        }
        if (tile + tstride < ntiles) ACX_LOAD_Y(tile + tstride)
        f32x16 acc[Cfg::kTiles];
#pragma unroll
        for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#pragma nounroll
        for (int k = 0; k < n; ++k) {
            const char* base1 = smem + (k == 0 ? 0 : 2 * k - 1) * Cfg::kSegBytes;
            const char* base2 = smem + (k == n - 1 ? 2 * n - 1 : 2 * k + 2) * Cfg::kSegBytes;
            f32x16 X[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 64 * k + 32 * j + 8 * q + 4 * hh);
                    X[j][4 * q + 0] = bq[0]; X[j][4 * q + 1] = bq[1]; X[j][4 * q + 2] = bq[2]; X[j][4 * q + 3] = bq[3];
                }
#pragma unroll
            for (int u = 0; u < Cfg::kUnits; ++u) {
                const f32x4 f = ACX_W1_RD(base1, u);
                X[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ACX_B8(f), ACX_B8(act[u >> 1]), X[u & 1], 0, 0, 0);
            }
            // GELU + bf16 pack: k-step s' = 2 j + (pair >> 2) of phase 2 takes pairs 4 (s' & 1) .. + 3 of X tile j
            f32x4 g[4];
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                unsigned un[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    GeluState3 gs;
                    const float ax = X[sp >> 1][2 * (4 * (sp & 1) + p)], ay = X[sp >> 1][2 * (4 * (sp & 1) + p) + 1];
                    gelu2h_micro<0>(gs, gk, ax, ay); gelu2h_micro<1>(gs, gk, ax, ay); gelu2h_micro<2>(gs, gk, ax, ay); gelu2h_micro<3>(gs, gk, ax, ay);
                    gelu2h_micro<4>(gs, gk, ax, ay); gelu2h_micro<5>(gs, gk, ax, ay); gelu2h_micro<6>(gs, gk, ax, ay);
                    un[p] = pack_bf16(gs.qx, gs.qy);
                }
                g[sp] = __builtin_bit_cast(f32x4, uint4{un[0], un[1], un[2], un[3]});
            }
#pragma unroll
            for (int u = 0; u < Cfg::kUnits; ++u) {
                const f32x4 f = ACX_W2_RD(base2, u);
                acc[u >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ACX_B8(f), ACX_B8(g[u & 3]), acc[u >> 2], 0, 0, 0);
            }
        }

        // (a tile whose 32 rows all lie inside M stores without a mask: behind `if (valid)` every store is a basic block of its
        // own and hipcc's waits for the residual values then also wait for the stores before them)
        auto epilogue = [&](auto masked) __attribute__((always_inline)) {
        constexpr bool kMasked = decltype(masked)::value;
        // ---- epilogue: lane (px, hh), tile t, q: channels 32 t + 8 q + 4 hh .. + 3  ->  x = x + out + b2 ----
        if constexpr (LNOUT) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 32 * t + 8 * q + 4 * hh);
                    const float4 v = xr[4 * t + q];
                    acc[t][4 * q + 0] += v.x + bb.x; acc[t][4 * q + 1] += v.y + bb.y;
                    acc[t][4 * q + 2] += v.z + bb.z; acc[t][4 * q + 3] += v.w + bb.w;
                    sum += (acc[t][4 * q + 0] + acc[t][4 * q + 1]) + (acc[t][4 * q + 2] + acc[t][4 * q + 3]);
                }
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / C);
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float u = acc[t][r] - mean; d = fmaf(u, u, d); }
            d += __shfl_xor(d, 32);
            const float rstd = 1.0f / sqrtf(d * (1.0f / C) + 1e-6f);
            {
                // 16-byte pieces by trading halves with the partner lane (see mlp_fused_wide_bf16_kernel)
                __bf16* op = ln_out + mrow * (long long)ld_out + 8 * hh;
#pragma unroll
                for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        unsigned e[2], o[2];
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            e[w] = pack_bf16((acc[t][8 * j + 2 * w] - mean) * rstd, (acc[t][8 * j + 2 * w + 1] - mean) * rstd);
                            o[w] = pack_bf16((acc[t][8 * j + 4 + 2 * w] - mean) * rstd, (acc[t][8 * j + 4 + 2 * w + 1] - mean) * rstd);
                            acx_pair_swap(e[w], o[w]);
                        }
                        if (!kMasked || valid) *reinterpret_cast<uint4*>(op + 32 * t + 16 * j) = uint4{e[0], e[1], o[0], o[1]};
                    }
                if (!kMasked || valid)
                    for (int c = C + 8 * hh; c < ld_out; c += 16)      // zero the K padding of the downsample GEMM's operand rows
                        *reinterpret_cast<uint4*>(ln_out + mrow * (long long)ld_out + c) = uint4{0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float4 v[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int q = 2 * j + i;
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 32 * t + 8 * q + 4 * hh);
                        v[i] = xr[4 * t + q];
                        v[i].x += acc[t][4 * q + 0] + bb.x;
                        v[i].y += acc[t][4 * q + 1] + bb.y;
                        v[i].z += acc[t][4 * q + 2] + bb.z;
                        v[i].w += acc[t][4 * q + 3] + bb.w;
                    }
                    if constexpr (ABF) {
                        unsigned e[2] = {acx_pack_bf16x2(v[0].x, v[0].y), acx_pack_bf16x2(v[0].z, v[0].w)};
                        unsigned o[2] = {acx_pack_bf16x2(v[1].x, v[1].y), acx_pack_bf16x2(v[1].z, v[1].w)};
                        acx_pair_swap(e[0], o[0]);
                        acx_pair_swap(e[1], o[1]);
                        if (!kMasked || valid)
                            *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(x) + mrow * C + 8 * hh + 32 * t + 16 * j) = uint4{e[0], e[1], o[0], o[1]};
                    } else if (!kMasked || valid) {
                        acx_st4<false>(x, mrow * C + 4 * hh + 32 * t + 16 * j, v[0]);
                        acx_st4<false>(x, mrow * C + 4 * hh + 32 * t + 16 * j + 8, v[1]);
                    }
                }
        }
        };
        if ((tile + 1) * 32 <= M) epilogue(std::false_type{});
        else epilogue(std::true_type{});
    }
#undef ACX_B8
#undef ACX_W1_RD
#undef ACX_W2_RD
#undef ACX_LOAD_Y
}

// number of CUs of the current device (persistent launches), cached per device
static int cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int v = cached[dev & 63].load(std::memory_order_acquire);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev & 63].store(v, std::memory_order_release);
    }
    return v;
}

template <bool LNOUT, bool ABF>
static int launch_stat_bf16(const BlockW& w, const void* y, void* x, long long M, void* ln_out, int ld_out, hipStream_t s) {
    static_assert(WideBfCfg<96, 1>::kSegs * WideBfCfg<96, 1>::kSegBytes + 5 * 96 * 4 <= kCuLdsBytes, "stream does not fit the LDS");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_stat_bf16_kernel<LNOUT, ABF>, kCuLdsBytes));
    const long long wgs_needed = ((M + 31) / 32 + 7) / 8;
    const long long blocks = wgs_needed < cu_count() ? wgs_needed : cu_count();
    launch_kernel(&mlp_fused_stat_bf16_kernel<LNOUT, ABF>, dim3((unsigned)blocks), dim3(512), kCuLdsBytes /* CU-exclusive */, s,
        y, x, reinterpret_cast<const char*>(w.wstream_b), w.b1, w.b2, M, ld_out, reinterpret_cast<__bf16*>(ln_out));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

bool mlp_fused_wide_bf16_supported(int C) { return C == 384 || C == 192 || C == 96; }

// LDS position of 16-byte chunk p of row `row` of a W1 image (acx_finalize packs the stream with it)
int mlp_fused_wide_bf16_swz(int C, int row) {
    return C == 384 ? WideBfCfg<384, 1>::swz1(row) : (C == 192 ? WideBfCfg<192, 1>::swz1(row) : WideBfCfg<96, 1>::swz1(row));
}

template <bool ABF>
static int launch_bf16_any(const BlockW& w, int C, const void* y, void* x, long long M, hipStream_t s, void* ln_out, int ld_out) {
    if (C == 384) return ln_out ? launch_wide_bf16_cfg<384, 1, true, ABF>(w, y, x, M, ln_out, ld_out, s) : launch_wide_bf16_cfg<384, 1, false, ABF>(w, y, x, M, nullptr, 0, s);
    // C = 192 with bf16 activations and no LayerNorm output: TWO pixel tiles per wave fit the register file without scratch, and
    // every weight fragment read from the LDS feeds two MFMAs instead of one (the ring kernels are LDS-bound at one): 220 ->
    // 208 us per block in the lab.  (fp32 activations: 204 B of scratch per lane; LNOUT: 140 B -- those stay at one tile.)
    if constexpr (ABF) { if (C == 192 && !ln_out) return launch_wide_bf16_cfg<192, 2, false, true>(w, y, x, M, nullptr, 0, s); }
    if (C == 192) return ln_out ? launch_wide_bf16_cfg<192, 1, true, ABF>(w, y, x, M, ln_out, ld_out, s) : launch_wide_bf16_cfg<192, 1, false, ABF>(w, y, x, M, nullptr, 0, s);
    if (C == 96) return ln_out ? launch_stat_bf16<true, ABF>(w, y, x, M, ln_out, ld_out, s) : launch_stat_bf16<false, ABF>(w, y, x, M, nullptr, 0, s);
    ACX_FAIL(ACX_ERR_SHAPE, "fused bf16 MLP: unsupported channel count %d", C);
}

int launch_mlp_fused_wide_bf16(acx_ctx* c, const BlockW& w, int C, const void* y, void* x, long long M, hipStream_t s,
                               void* ln_out, int ld_out, bool act_bf16) {
    if (!w.wstream_b) ACX_FAIL(ACX_ERR_STATE, "fused bf16 MLP: the weight stream was not packed for C=%d", C);
    ProfScope ps(c, ACX_K_MLP_WIDE, s);
    return act_bf16 ? launch_bf16_any<true>(w, C, y, x, M, s, ln_out, ld_out) : launch_bf16_any<false>(w, C, y, x, M, s, ln_out, ld_out);
}

}  // namespace acx
