// K4ps -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86) in the
// default fp32_split arithmetic (fp32 operands as fp16 hi + lo, three fp16 MFMAs per product, fp32 accumulate -- DESIGN.md 3a) as
// PRODUCER / CONSUMER wave pairs, for the narrow stages (C = 96, 192).  Round 5.
//
// The dataflow is mlp_pair_bf16.hip's (see its header): a persistent CU-exclusive workgroup of eight waves, two per SIMD with
// different jobs -- the producer (waves 0-3) owns the LayerNorm'ed rows of 32 PT pixels as S16 halves, computes
// X = W1c . LN(y)^T for one chunk of 32 hidden units per interval and turns the previous chunk into G = GELU(X) as fp16 hi / lo
// in the LDS; the consumer (waves 4-7) owns the out accumulators of the same pixels and runs out^T += W2c . G two intervals
// later.  One workgroup barrier per interval; weights as a stream of 32-hidden-unit S16 segments (128 C bytes) in consumption
// order through a 4-slot LDS ring, their LDS-DMA pieces dealt over all eight waves and threaded through the MFMA loops.
// With three MFMAs per fragment pair an interval carries 3 x (24 PT + 24 PT) x (C / 192) matrix instructions against the same
// fixed costs (barrier, pieces, fragment reads) as in bf16: the lab form of the bf16 kernel with tripled MFMAs ran at 0.83 of
// the matrix floor inside an interval (profiles/r05_d_pair_triple.txt) where the one-wave-per-SIMD kernels of
// mlp_fused_wide.hip / mlp_fused_split.hip reach 0.5.
// Matrix instruction: v_mfma_f32_32x32x16_f16 -- lane (l31, hh): row / column l31, k half hh -- so that every layout is the bf16
// pair kernel's; operands in S16 form: per 8 k values a 32-byte block [8 x hi][8 x lo] (two 16-byte chunks).
// Arithmetic and rounding points are those of the other fp32_split kernels: LayerNorm rows x 2^11 and weights x 2^s (per
// layer) split into fp16 hi + lo, products hi.hi + hi.lo + lo.hi in fp32, GELU in fp32 (split_math.h, third form) scaled by
// the block's 2^h and split again, the epilogue multiplies by the exact inverse scales.
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {

template <int C, int PT>
struct PairSplitCfg {
    static constexpr int kHC = 32;                          // hidden units per chunk
    static constexpr int kChunks = 4 * C / kHC;             // n
    static constexpr int kSegBytes = kHC * C * 4;           // [32][C] W1 image or [C][32] W2 image, S16 (4 bytes per element)
    static constexpr int kSegPieces = kSegBytes / 1024;
    static constexpr int kSteps = C / 16;                   // k-steps of phase 1
    static constexpr int kTiles = C / 32;                   // out-channel tiles of phase 2
    static constexpr int kPairPix = 32 * PT;
    static constexpr int kPix = 4 * kPairPix;               // pixels of a workgroup tile
    static constexpr int kGBytes = PT * 2 * 2 * 1024;       // G of one chunk of one pair: PT x 2 k-steps x (hi, lo) x 64 lanes x 16 B
    static constexpr int kOffG = 4 * kSegBytes;             // LDS: ring (4 slots) | G [4 pairs][2 slots] | b1 [4C] (scaled) | b2 [C]
    static constexpr int kOffB1 = kOffG + 4 * 2 * kGBytes;
    static constexpr int kOffB2 = kOffB1 + 4 * C * 4;
    static constexpr size_t kLdsBytes = (size_t)kOffB2 + C * 4;
    static_assert(kChunks % 4 == 0 && kSegPieces % 4 == 0, "ring slots are static per interval parity; pieces dealt over eight waves");
    // W1 rows are 4 C bytes = C/4 chunks of 16 B: the XOR that spreads consecutive rows over the LDS banks
    static constexpr int kSwzBits = ((4 * C) % 256 == 0) ? 4 : 3;
    __host__ __device__ static int swz1(int row) { return kSwzBits == 4 ? (row & 15) : ((row >> 1) & 7); }
    // stream position of the segments (consumption order, period 2n): W1(k) is read in interval k, W2(j) in interval j + 2
    __host__ __device__ static constexpr int pos_w1(int k) { return k == 0 ? 0 : (k == 1 ? 2 : 2 * k - 1); }
    __host__ __device__ static constexpr int pos_w2(int j) { return j == kChunks - 1 ? 1 : (j == kChunks - 2 ? 2 * kChunks - 1 : 2 * j + 4); }
};

template <int I, int N, class F>
__device__ __forceinline__ void acx_ps_for_impl(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); acx_ps_for_impl<I + 1, N>(f); }
}
template <int N, class F>
__device__ __forceinline__ void acx_ps_for(F&& f) { acx_ps_for_impl<0, N>(f); }

template <int C, int PT, bool LNOUT>
__global__ __launch_bounds__(512) void mlp_pair_split_kernel(
    const float* __restrict__ y, float* __restrict__ x, const char* __restrict__ wstream /*[2n][128 C bytes], position order*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, float sinv1, float sinv2, float hscale,
    char* __restrict__ ln_out /* LNOUT: (M, C) S16 rows of LayerNorm(x_new) x 2^11, written INSTEAD of x */) {
    using Cfg = PairSplitCfg<C, PT>;
    constexpr int n = Cfg::kChunks;
    constexpr int SEG = Cfg::kSegBytes;
    constexpr int P = Cfg::kSegPieces;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + Cfg::kOffB1);     // [4C], pre-divided by sinv1 (accumulator units)
    float* b2s = reinterpret_cast<float*>(smem + Cfg::kOffB2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    ACX_CLAIM_VGPR(255);          // CU-exclusive: 8 waves x 256 registers hold the SIMDs' whole register files

    const unsigned smem_a = acx_lds_addr(smem);
    // LDS-DMA piece i of this wave: q = wave + 8 i of the 2 P pieces of the next interval's two segments (an interval that needs
    // one segment passes it twice: same bytes into the same place -- no conditional around an MFMA loop)
    auto dma_piece = [&](const int i, const int pos0, const int pos1) __attribute__((always_inline)) {
        const int q = wave + 8 * i;
        const int second = q >= P ? 1 : 0;
        const int piece = q - second * P;
        const int pos = second ? pos1 : pos0;
        const char* src = wstream + (long long)pos * SEG + piece * 1024 + lane * 16;
        acx_glds16_own_m0(src, smem_a + (unsigned)((pos & 3) * SEG + piece * 1024));
    };
    constexpr int kCnt = 2 * P / 8;                 // pieces per wave and interval
    auto request = [&](const int pos0, const int pos1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kCnt; ++i) dma_piece(i, pos0, pos1);
    };
#define ACX_ENDINT { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

    {
        const float b1scale = 1.0f / sinv1;         // a power of two
        for (int i = tid; i < 4 * C; i += 512) b1s[i] = b1[i] * b1scale;
        if (tid < C) b2s[tid] = b2[tid];
    }
    request(0, 1);                                  // interval 0 of the first tile
    ACX_ENDINT

    const long long ntiles = (M + Cfg::kPix - 1) / Cfg::kPix;
    constexpr int kD = 3;                           // rotating fragment registers: reads run two units ahead of their MFMAs
#define ACX_H8(v_) __builtin_bit_cast(h8, v_)
#define ACX_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)

    if (wave < 4) {
        // ================================ producer ================================
        // W1 image: row r = hidden unit (4 C bytes = C/4 chunks of 16 B); k-step s, half hh, part pl (0 hi, 1 lo) = chunk
        // 4 s + 2 hh + pl at position ^ swz1(r): the XOR touches the low kSwzBits bits
        constexpr int kVarS = Cfg::kSwzBits == 4 ? 4 : 2;            // k-steps whose chunk index differs in the XORed bits
        int w1off[kVarS][2];
#pragma unroll
        for (int q = 0; q < kVarS; ++q)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w1off[q][pl] = l31 * (4 * C) + (((4 * q + 2 * hh + pl) ^ Cfg::swz1(l31)) << 4);
        GeluK3 gk = gelu_k3(sinv1, hscale);
        gelu_k3_to_vgprs(gk);                       // two waves per SIMD: a scalar operand costs a vector instruction 2 extra cycles
        f32x4 acth[PT][Cfg::kSteps], actl[PT][Cfg::kSteps];     // lane (px = l31, half hh): channels 16 s + 8 hh .. + 7 as 8 fp16 hi / lo
        f32x16 Xa[PT], Xb[PT];                      // pre-activation tiles: one accumulates while the other's GELU is evaluated
        char* gbase = smem + Cfg::kOffG + pair * 2 * Cfg::kGBytes;

        // LayerNorm of the rows of a tile (fp32 statistics: mean, then the centred sum of squares) x 2^11 -> S16 halves
        constexpr int kEarlyY = (2 * Cfg::kSteps) / 3 / PT;      // k-steps per pixel tile requested ahead: 64 registers
        f32x4 yraw[PT][kEarlyY][2];
        auto prefetch_y = [&](const long long tile) __attribute__((always_inline)) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                long long r = tile * Cfg::kPix + pair * Cfg::kPairPix + pt * 32 + l31;
                if (r >= M) r = M - 1;
                const float* yp = y + r * C + 8 * hh;
#pragma unroll
                for (int s = 0; s < kEarlyY; ++s) { yraw[pt][s][0] = *reinterpret_cast<const f32x4*>(yp + 16 * s); yraw[pt][s][1] = *reinterpret_cast<const f32x4*>(yp + 16 * s + 4); }
            }
        };
        auto load_ln = [&](const long long tile, auto pre_tag) __attribute__((always_inline)) {
            constexpr bool kPre = decltype(pre_tag)::value;       // the first kEarlyY k-steps are in yraw already
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                long long r = tile * Cfg::kPix + pair * Cfg::kPairPix + pt * 32 + l31;
                if (r >= M) r = M - 1;
                const float* yp = y + r * C + 8 * hh;
                float a[C / 2];
#pragma unroll
                for (int s = 0; s < Cfg::kSteps; ++s) {
                    f32x4 v0, v1;
                    if (kPre && s < kEarlyY) { v0 = yraw[pt][s][0]; v1 = yraw[pt][s][1]; }
                    else { v0 = *reinterpret_cast<const f32x4*>(yp + 16 * s); v1 = *reinterpret_cast<const f32x4*>(yp + 16 * s + 4); }
                    a[8 * s] = v0[0]; a[8 * s + 1] = v0[1]; a[8 * s + 2] = v0[2]; a[8 * s + 3] = v0[3];
                    a[8 * s + 4] = v1[0]; a[8 * s + 5] = v1[1]; a[8 * s + 6] = v1[2]; a[8 * s + 7] = v1[3];
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
                    for (int p = 0; p < 4; ++p) acx_split_pair((a[8 * s + 2 * p] - mean) * sc, (a[8 * s + 2 * p + 1] - mean) * sc, uh[p], ul[p]);
                    acth[pt][s] = __builtin_bit_cast(f32x4, uint4{uh[0], uh[1], uh[2], uh[3]});
                    actl[pt][s] = __builtin_bit_cast(f32x4, uint4{ul[0], ul[1], ul[2], ul[3]});
                }
            }
        };
        // G of a chunk: registers 8 sp .. 8 sp + 7 of a pixel tile's accumulator are this lane's 8 k values of phase 2's k-step sp
        unsigned unh[PT][8], unl[PT][8];
        GeluState3 gst;
        // nano-step sg of the 168 PT that turn Xv into (unh, unl): pixel tile sg / 168, register pair (sg % 168) / 21, step sg % 21
#define ACX_NANO(Xv_, sg_)                                                                                      \
        {   constexpr int mt_ = (sg_) / (8 * kGelu3Nano), pr_ = ((sg_) % (8 * kGelu3Nano)) / kGelu3Nano, st_ = (sg_) % kGelu3Nano; \
            gelu3_nano<st_>(gst, gk, Xv_[mt_][2 * pr_], Xv_[mt_][2 * pr_ + 1], unh[mt_][pr_], unl[mt_][pr_]); }
        auto write_g = [&](const int kc) __attribute__((always_inline)) {
            char* gs = gbase + (kc & 1) * Cfg::kGBytes + lane * 16;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    *reinterpret_cast<f32x4*>(gs + ((pt * 2 + sp) * 2 + 0) * 1024) = __builtin_bit_cast(f32x4, uint4{unh[pt][4 * sp], unh[pt][4 * sp + 1], unh[pt][4 * sp + 2], unh[pt][4 * sp + 3]});
                    *reinterpret_cast<f32x4*>(gs + ((pt * 2 + sp) * 2 + 1) * 1024) = __builtin_bit_cast(f32x4, uint4{unl[pt][4 * sp], unl[pt][4 * sp + 1], unl[pt][4 * sp + 2], unl[pt][4 * sp + 3]});
                }
        };
        // One producer interval: Xn = b1 + W1c(kc) . LN(y)^T from ring slot `slot` (three MFMAs per k-step and pixel tile); behind
        // the MFMAs of every k-step ride this step's share of the GELU + split of Xv (chunk kc - 1; HV: there is one) and of the
        // wave's LDS-DMA pieces (next interval's segments rq0, rq1; in the first two thirds of the loop); G(kc - 1) leaves at the end
        auto phase1 = [&](auto hv_tag, f32x16 (&Xn)[PT], f32x16 (&Xv)[PT], const int kc, const int slot, const int rq0, const int rq1) __attribute__((always_inline)) {
            constexpr bool HV = decltype(hv_tag)::value;
            constexpr int kT = 8 * kGelu3Nano * PT, kS = Cfg::kSteps, kSpan = (2 * kS) / 3;
            const char* base = smem + slot * SEG;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * kc + 8 * q + 4 * hh);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) { Xn[pt][4 * q] = bq[0]; Xn[pt][4 * q + 1] = bq[1]; Xn[pt][4 * q + 2] = bq[2]; Xn[pt][4 * q + 3] = bq[3]; }
            }
#define ACX_W1_RD(s_, pl_) (*reinterpret_cast<const f32x4*>(base + ((s_) / kVarS) * (kVarS * 64) + w1off[(s_) % kVarS][pl_]))
            f32x4 fh[kD], fl[kD];
#pragma unroll
            for (int i = 0; i < kD - 1; ++i) { fh[i] = ACX_W1_RD(i, 0); fl[i] = ACX_W1_RD(i, 1); }
            acx_ps_for<kS>([&](auto s_tag) __attribute__((always_inline)) {
                constexpr int sx = decltype(s_tag)::value;
                if constexpr (sx + kD - 1 < kS) { fh[(sx + kD - 1) % kD] = ACX_W1_RD(sx + kD - 1, 0); fl[(sx + kD - 1) % kD] = ACX_W1_RD(sx + kD - 1, 1); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    Xn[pt] = ACX_MFMA(ACX_H8(fh[sx % kD]), ACX_H8(acth[pt][sx]), Xn[pt]);
                    Xn[pt] = ACX_MFMA(ACX_H8(fh[sx % kD]), ACX_H8(actl[pt][sx]), Xn[pt]);
                    Xn[pt] = ACX_MFMA(ACX_H8(fl[sx % kD]), ACX_H8(acth[pt][sx]), Xn[pt]);
                }
                if constexpr (sx < kSpan) {
#pragma unroll
                    for (int i = sx * kCnt / kSpan; i < (sx + 1) * kCnt / kSpan; ++i) dma_piece(i, rq0, rq1);
                }
                if constexpr (HV)
                    acx_ps_for<kT * (sx + 1) / kS - kT * sx / kS>([&](auto g_tag) __attribute__((always_inline)) {
                        ACX_NANO(Xv, kT * sx / kS + decltype(g_tag)::value)
                    });
                __builtin_amdgcn_sched_barrier(0);
            });
#undef ACX_W1_RD
            if constexpr (HV) write_g(kc - 1);
        };

        // the first kEarlyY k-steps of the NEXT tile's rows are requested at the top of interval n - 1 (they arrive while it runs: its
        // closing wait covers them), the rest at interval n behind the last GELU -- registers for all of them are not to be had
        long long tile = blockIdx.x;
        if (tile < ntiles) load_ln(tile, std::false_type{});
        for (; tile < ntiles; tile += gridDim.x) {
            const bool more = tile + gridDim.x < ntiles;
            // k = 0 (requests for interval 1: W1(1))
            phase1(std::false_type{}, Xa, Xb, 0, Cfg::pos_w1(0) & 3, Cfg::pos_w1(1), Cfg::pos_w1(1));
            ACX_ENDINT
            // k = 1 (requests for interval 2: W1(2), W2(0))
            phase1(std::true_type{}, Xb, Xa, 1, Cfg::pos_w1(1) & 3, Cfg::pos_w1(2), Cfg::pos_w2(0));
            ACX_ENDINT
            // k = 2 .. n - 1, two per trip: W1 in slot 3 (k even) / 1 (k odd); requests for interval k + 1: W1(k + 1), W2(k - 1)
            // (interval n reads W2(n - 2) only)
#pragma nounroll
            for (int k = 2; k < n - 2; k += 2) {
                phase1(std::true_type{}, Xa, Xb, k, 3, 2 * k + 1, 2 * k + 2);
                ACX_ENDINT
                phase1(std::true_type{}, Xb, Xa, k + 1, 1, 2 * k + 3, 2 * k + 4);
                ACX_ENDINT
            }
            phase1(std::true_type{}, Xa, Xb, n - 2, 3, 2 * n - 3, 2 * n - 2);
            ACX_ENDINT
            prefetch_y(more ? tile + gridDim.x : tile);
            phase1(std::true_type{}, Xb, Xa, n - 1, 1, 2 * n - 1, 2 * n - 1);       // (interval n reads W2(n - 2) only)
            ACX_ENDINT
            // k = n: the last GELU (chunk n - 1 sits in Xb: n - 1 is odd); requests for the next tile's interval 0; the next tile's rows
            request(0, 1);
            acx_ps_for<8 * kGelu3Nano * PT>([&](auto g_tag) __attribute__((always_inline)) { ACX_NANO(Xb, decltype(g_tag)::value) });
            write_g(n - 1);
            if (more) load_ln(tile + gridDim.x, std::true_type{});
            ACX_ENDINT
        }
#undef ACX_NANO
        // the consumer's two intervals behind the last tile
        ACX_ENDINT
        ACX_ENDINT
    } else {
        // ================================ consumer ================================
        // W2 image: row = out channel (128 B = 8 chunks); k-step sp, half hh, part pl = chunk 4 sp + 2 hh + pl at position ^ acx_swz8(row)
        int w2off[2][2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w2off[sp][pl] = l31 * 128 + (((4 * sp + 2 * hh + pl) ^ acx_swz8(l31)) << 4);
        f32x16 acc[PT][Cfg::kTiles];
        const char* gbase = smem + Cfg::kOffG + pair * 2 * Cfg::kGBytes + lane * 16;

        // out^T += W2c(chunk j) . G(j): the segment sits in ring slot `slot`, G in slot j & 1; this wave's LDS-DMA pieces of the next
        // interval's segments (rq0, rq1) go out between the MFMAs in the first two thirds of the loop.  ZERO: the accumulators start here.
        auto phase2 = [&](auto zero_tag, const int j, const int slot, const int rq0, const int rq1) __attribute__((always_inline)) {
            constexpr bool kZero = decltype(zero_tag)::value;
            const char* base = smem + slot * SEG;
            constexpr int kUnits = 2 * Cfg::kTiles;
            constexpr int kSpan = (2 * kUnits) / 3;
            f32x4 gh[PT][2], gl[PT][2];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    gh[pt][sp] = *reinterpret_cast<const f32x4*>(gbase + (j & 1) * Cfg::kGBytes + ((pt * 2 + sp) * 2 + 0) * 1024);
                    gl[pt][sp] = *reinterpret_cast<const f32x4*>(gbase + (j & 1) * Cfg::kGBytes + ((pt * 2 + sp) * 2 + 1) * 1024);
                }
            // unit u = (out tile t = u >> 1, k-step sp = u & 1)
#define ACX_W2_RD(u_, pl_) (*reinterpret_cast<const f32x4*>(base + ((u_) >> 1) * 4096 + w2off[(u_) & 1][pl_]))
            f32x4 fh[kD], fl[kD];
#pragma unroll
            for (int i = 0; i < kD - 1; ++i) { fh[i] = ACX_W2_RD(i, 0); fl[i] = ACX_W2_RD(i, 1); }
#pragma unroll
            for (int u = 0; u < kUnits; ++u) {
                if (u + kD - 1 < kUnits) { fh[(u + kD - 1) % kD] = ACX_W2_RD(u + kD - 1, 0); fl[(u + kD - 1) % kD] = ACX_W2_RD(u + kD - 1, 1); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    if (kZero && (u & 1) == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[pt][u >> 1][i] = 0.f;
                    }
                    acc[pt][u >> 1] = ACX_MFMA(ACX_H8(fh[u % kD]), ACX_H8(gh[pt][u & 1]), acc[pt][u >> 1]);
                    acc[pt][u >> 1] = ACX_MFMA(ACX_H8(fh[u % kD]), ACX_H8(gl[pt][u & 1]), acc[pt][u >> 1]);
                    acc[pt][u >> 1] = ACX_MFMA(ACX_H8(fl[u % kD]), ACX_H8(gh[pt][u & 1]), acc[pt][u >> 1]);
                }
#pragma unroll
                for (int i = (u < kSpan ? u * kCnt / kSpan : kCnt); i < (u + 1 < kSpan ? (u + 1) * kCnt / kSpan : kCnt) && u < kSpan; ++i)
                    dma_piece(i, rq0, rq1);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef ACX_W2_RD
        };
        // the residual rows of a tile, in the accumulator layout: lane (px, hh), tile t, q: channels 32 t + 8 q + 4 hh .. + 3
        f32x4 xr[PT][4 * Cfg::kTiles];
        auto load_x = [&](const long long tile) __attribute__((always_inline)) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                long long r = tile * Cfg::kPix + pair * Cfg::kPairPix + pt * 32 + l31;
                if (r >= M) r = M - 1;
                const float* xp = x + r * C + 4 * hh;
#pragma unroll
                for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xr[pt][4 * t + q] = *reinterpret_cast<const f32x4*>(xp + 32 * t + 8 * q);
            }
        };
        // x + out x sinv2 + b2 -> x, or its LayerNorm x 2^11 -> the downsample GEMM's S16 operand rows
        auto epilogue = [&](const long long tile, auto masked_tag) __attribute__((always_inline)) {
            constexpr bool kMasked = decltype(masked_tag)::value;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const long long r = tile * Cfg::kPix + pair * Cfg::kPairPix + pt * 32 + l31;
                const bool valid = !kMasked || r < M;
                const long long rr = r < M ? r : M - 1;
                if constexpr (LNOUT) {
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 32 * t + 8 * q + 4 * hh);
                            const f32x4 v = xr[pt][4 * t + q];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[pt][t][4 * q + e] = v[e] + fmaf(acc[pt][t][4 * q + e], sinv2, bb[e]);
                            sum += (acc[pt][t][4 * q] + acc[pt][t][4 * q + 1]) + (acc[pt][t][4 * q + 2] + acc[pt][t][4 * q + 3]);
                        }
                    sum += __shfl_xor(sum, 32);
                    const float mean = sum * (1.0f / C);
                    float d = 0.f;
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) { const float u = acc[pt][t][i] - mean; d = fmaf(u, u, d); }
                    d += __shfl_xor(d, 32);
                    const float sc = kSplitLnScale / sqrtf(d * (1.0f / C) + 1e-6f);
                    // an S16 block (8 channels: [8 hi][8 lo]) is shared by the lanes (px, 0) and (px, 1): after a permlane32 swap the
                    // lower lane holds all 8 hi halves and the upper lane all 8 lo halves -> one 16-byte store each
                    char* op = ln_out + rr * (long long)(C * 4) + 16 * hh;
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            unsigned uhi[2], ulo[2];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                acx_split_pair((acc[pt][t][4 * q + 2 * e] - mean) * sc, (acc[pt][t][4 * q + 2 * e + 1] - mean) * sc, uhi[e], ulo[e]);
                                acx_pair_swap(uhi[e], ulo[e]);
                            }
                            if (valid) *reinterpret_cast<uint4*>(op + (4 * t + q) * 32) = uint4{uhi[0], uhi[1], ulo[0], ulo[1]};
                        }
                } else if (valid) {
                    float* xp = x + rr * C + 4 * hh;
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 32 * t + 8 * q + 4 * hh);
                            f32x4 v = xr[pt][4 * t + q];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += fmaf(acc[pt][t][4 * q + e], sinv2, bb[e]);
                            *reinterpret_cast<f32x4*>(xp + 32 * t + 8 * q) = v;
                        }
                }
            }
        };
        // The epilogue's stores and the residual loads stay in flight across the barrier of their interval: they are the wave's youngest
        // operations (issued behind its LDS-DMA pieces), so a counted wait that leaves exactly them outstanding still covers the pieces.
        constexpr int kRowOps = 4 * Cfg::kTiles * PT;          // 16-byte stores of an epilogue = 16-byte loads of load_x
        static_assert(kRowOps < 64, "vmcnt is a 6-bit counter");
#define ACX_ENDINT_KEEP(n_) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(n_) : "memory"); __builtin_amdgcn_sched_barrier(0); }
        auto leave = [&](const long long t) __attribute__((always_inline)) {      // the tile's results leave; ends the interval
            if ((t + 1) * Cfg::kPix <= M) {
                epilogue(t, std::false_type{});
                ACX_ENDINT_KEEP(kRowOps)
            } else {
                epilogue(t, std::true_type{});
                ACX_ENDINT
            }
        };

        long long prev = -1;
        for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            // k = 0: the previous tile's last chunk (first tile: the same instructions on whatever the registers hold -- nothing reads
            // the result; no branch around an MFMA loop), then its results leave; requests for interval 1: W1(1)
            phase2(std::false_type{}, n - 1, Cfg::pos_w2(n - 1) & 3, Cfg::pos_w1(1), Cfg::pos_w1(1));
            if (prev >= 0) leave(prev);
            else ACX_ENDINT
            // k = 1: this tile's residual rows are requested (they have the whole tile to arrive); requests for interval 2: W1(2), W2(0)
            request(Cfg::pos_w1(2), Cfg::pos_w2(0));
            load_x(tile);
            ACX_ENDINT_KEEP(kRowOps)
            // k = 2: the accumulators start; W2(k - 2) sits in slot 0 (k even) / 2 (k odd); requests for interval k + 1: W1(k + 1), W2(k - 1)
            phase2(std::true_type{}, 0, 0, 5, 6);
            ACX_ENDINT
            phase2(std::false_type{}, 1, 2, n == 4 ? 2 * n - 1 : 7, n == 4 ? 2 * n - 1 : 8);
            ACX_ENDINT
#pragma nounroll
            for (int k = 4; k < n; k += 2) {
                phase2(std::false_type{}, k - 2, 0, 2 * k + 1, 2 * k + 2);
                ACX_ENDINT
                const bool last = k + 1 == n - 1;
                phase2(std::false_type{}, k - 1, 2, last ? 2 * n - 1 : 2 * k + 3, last ? 2 * n - 1 : 2 * k + 4);
                ACX_ENDINT
            }
            // k = n; requests for the next tile's interval 0: W1(0), W2(n - 1)
            phase2(std::false_type{}, n - 2, Cfg::pos_w2(n - 2) & 3, 0, 1);
            ACX_ENDINT
            prev = tile;
        }
        phase2(std::false_type{}, n - 1, Cfg::pos_w2(n - 1) & 3, 0, 1);      // (its requests are idle: nothing reads them)
        leave(prev);
        ACX_ENDINT
#undef ACX_ENDINT_KEEP
    }
#undef ACX_H8
#undef ACX_MFMA
#undef ACX_ENDINT
}

static int pair_split_cu_count() {
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

template <int C, int PT, bool LNOUT>
static int launch_pair_split_cfg(const BlockW& w, const float* y, float* x, long long M, void* ln_out, hipStream_t s) {
    using Cfg = PairSplitCfg<C, PT>;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "ring + G + biases do not fit the LDS");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_pair_split_kernel<C, PT, LNOUT>, kCuLdsBytes));
    const long long tiles = (M + Cfg::kPix - 1) / Cfg::kPix;
    long long share = pair_split_cu_count() / inflight_ways();       // one persistent workgroup per CU of this launch's share
    if (share < 1) share = 1;
    const long long blocks = tiles < share ? tiles : share;
    launch_kernel(&mlp_pair_split_kernel<C, PT, LNOUT>, dim3((unsigned)blocks), dim3(512), kCuLdsBytes /* CU-exclusive */, s,
        y, x, reinterpret_cast<const char*>(w.wstream_ps), w.b1, w.b2, M, 1.0f / (kSplitLnScale * w.w1s_scale),
        1.0f / (w.hid_scale * w.w2s_scale), w.hid_scale, reinterpret_cast<char*>(ln_out));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

#ifndef ACX_PS_PT96
#define ACX_PS_PT96 2
#endif
bool mlp_pair_split_supported(int C) { return C == 192 || C == 96; }
int mlp_pair_split_swz(int C, int row) { return C == 192 ? PairSplitCfg<192, 1>::swz1(row) : PairSplitCfg<96, ACX_PS_PT96>::swz1(row); }
int mlp_pair_split_pos_w1(int C, int k) { return C == 192 ? PairSplitCfg<192, 1>::pos_w1(k) : PairSplitCfg<96, ACX_PS_PT96>::pos_w1(k); }
int mlp_pair_split_pos_w2(int C, int j) { return C == 192 ? PairSplitCfg<192, 1>::pos_w2(j) : PairSplitCfg<96, ACX_PS_PT96>::pos_w2(j); }

int launch_mlp_pair_split(acx_ctx* c, const BlockW& w, int C, const float* y, float* x, long long M, hipStream_t s, void* ln_out) {
    if (!w.wstream_ps) ACX_FAIL(ACX_ERR_STATE, "paired split MLP: the weight stream was not packed for C=%d", C);
    ProfScope ps(c, C == 96 ? ACX_K_MLP_FUSED : ACX_K_MLP_WIDE, s);
    if (C == 192) return ln_out ? launch_pair_split_cfg<192, 1, true>(w, y, x, M, ln_out, s) : launch_pair_split_cfg<192, 1, false>(w, y, x, M, nullptr, s);
    if (C == 96) return ln_out ? launch_pair_split_cfg<96, ACX_PS_PT96, true>(w, y, x, M, ln_out, s) : launch_pair_split_cfg<96, ACX_PS_PT96, false>(w, y, x, M, nullptr, s);
    ACX_FAIL(ACX_ERR_SHAPE, "paired split MLP: unsupported channel count %d", C);
}

}  // namespace acx
