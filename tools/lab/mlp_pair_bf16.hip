// K4p -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86) in the bf16
// arithmetics, as PRODUCER / CONSUMER wave pairs (round 5; VERDICT r04 item 1: "two instruction streams per SIMD").
//
// The ring kernels of mlp_fused_wide_bf16.hip run ONE wave per SIMD that has to carry everything -- MFMAs, the GELU, 48
// fragment reads, 12 LDS-DMA pieces and a barrier per 48 MFMAs -- and every LDS-DMA piece stalls its issue for 60-180 cycles
// with nobody to cover it (0.31 of the bf16 matrix peak).  Here a persistent, CU-exclusive workgroup of EIGHT waves holds two
// waves per SIMD with different jobs (waves w and w + 4 share a SIMD):
//   * the producer (waves 0-3) owns the LayerNorm'ed activations of 32 PT pixels (B operand of phase 1, C/4 registers per
//     pixel tile), computes X = W1c . LN(y)^T for one chunk of 32 hidden units per interval, evaluates the GELU of the previous
//     chunk and leaves G = bf16(GELU(X)) in the LDS -- in the lane order in which it is phase 2's B operand (W2c's columns are
//     packed in that order, api.hip);
//   * the consumer (waves 4-7) owns the out accumulators of the same pixels (C/2 registers per pixel tile), initialised with
//     the residual x + b2, reads G two intervals later and runs out^T += W2c . G; its epilogue rounds and stores, and requests
//     the next tile's residual rows.
// An interval ends with one workgroup barrier.  Inside an interval the producer does its vector work FIRST (GELU) while the
// consumer's MFMAs have the matrix pipe, then its own MFMAs: matrix beside vector on every SIMD, and whichever wave stalls on an
// LDS-DMA issue, an LDS read or a wait is covered by its partner.  Weights arrive as a stream of 32-hidden-unit segments in the
// order of consumption (W1(k) and W2(k - 2) in interval k) through a 4-slot LDS ring, requested one interval ahead by all
// eight waves (LDS-DMA from inline asm, split_math.h).  Tiles follow each other without a drain: the consumer's last two
// intervals of a tile are the producer's first two of the next one.
// Pixels per pair: 32 PT; hidden chunk: 32; per interval and SIMD 24 PT + 24 PT MFMAs (C = 384: PT = 1, C = 192: PT = 2).
// Rounding points are those of mlp_fused_wide_bf16.hip (LayerNorm output and GELU output to bf16, weights at acx_finalize);
// the residual enters the accumulator first instead of last (fp32 either way).
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

// Round 6: this kernel left libacx.so (measured no faster in the forward); it builds only inside
// tools/lab/pair_lab.hip.  The lab hands its stream in through BlockW::wstream_b (the packing of this kernel -- 32-hidden-unit
// segments at pos_w1 / pos_w2 -- is in the git history of api.hip, round 5).
namespace acx {
bool mlp_pair_bf16_supported(int C);
int mlp_pair_bf16_swz(int C, int row);
int mlp_pair_bf16_pos_w1(int C, int k);
int mlp_pair_bf16_pos_w2(int C, int j);
int launch_mlp_pair_bf16(acx_ctx* c, const BlockW& w, int C, const void* y, void* x, long long M, hipStream_t s,
                         void* ln_out, int ld_out, bool act_bf16);
}

namespace acx {

typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));

template <int C, int PT>
struct PairCfg {
    static constexpr int kHC = 32;                          // hidden units per chunk
    static constexpr int kChunks = 4 * C / kHC;             // n
    static constexpr int kSegBytes = kHC * C * 2;           // [32][C] W1 image or [C][32] W2 image, bf16
    static constexpr int kSegPieces = kSegBytes / 1024;
    static constexpr int kRing = 4;
    static constexpr int kSteps = C / 16;                   // k-steps of phase 1
    static constexpr int kTiles = C / 32;                   // out-channel tiles of phase 2
    static constexpr int kPairPix = 32 * PT;
    static constexpr int kPix = 4 * kPairPix;               // pixels of a workgroup tile
    static constexpr int kGBytes = PT * 2 * 1024;           // G of one chunk of one pair: PT x 2 k-steps x 64 lanes x 16 B
    static constexpr int kOffG = kRing * kSegBytes;         // LDS: ring | G [4 pairs][2 slots] | b1 [4C] | b2 [C]
    static constexpr int kOffB1 = kOffG + 4 * 2 * kGBytes;
    static constexpr int kOffB2 = kOffB1 + 4 * C * 4;
    static constexpr size_t kLdsBytes = (size_t)kOffB2 + C * 4;
    static_assert(kChunks % 4 == 0 && kSegBytes % 1024 == 0, "ring slots are static per interval parity");
    // W1 rows are 2 C bytes = C/8 chunks of 16 B: the XOR that spreads 16 consecutive rows over the LDS banks (as WideBfCfg)
    static constexpr int kSwzBits = (C % 128 == 0) ? 4 : ((C % 64 == 0) ? 3 : 2);
    __host__ __device__ static int swz1(int row) { return kSwzBits == 4 ? (row & 15) : (kSwzBits == 3 ? ((row >> 1) & 7) : ((row >> 2) & 3)); }
    // stream position of the segments (consumption order, period 2n): W1(k) is read in interval k, W2(j) in interval j + 2
    __host__ __device__ static constexpr int pos_w1(int k) { return k == 0 ? 0 : (k == 1 ? 2 : 2 * k - 1); }
    __host__ __device__ static constexpr int pos_w2(int j) { return j == kChunks - 1 ? 1 : (j == kChunks - 2 ? 2 * kChunks - 1 : 2 * j + 4); }
};

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression in the body
template <int I, int N, class F>
__device__ __forceinline__ void acx_static_for_impl(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); acx_static_for_impl<I + 1, N>(f); }
}
template <int N, class F>
__device__ __forceinline__ void acx_static_for(F&& f) { acx_static_for_impl<0, N>(f); }

__device__ __forceinline__ unsigned pair_pack_bf16(float a, float b) {
    f32x2 v; v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2p));
}

#ifndef ACX_PAIR_STAMP_SKIP
#define ACX_PAIR_STAMP_SKIP 1
#endif
#ifdef ACX_PAIR_STAMPS      // lab builds only (tools/lab/pair_lab.hip): s_memtime at the marks of the first tile, waves 0 and 4
constexpr int kPairStampBlocks = 64, kPairStampSlots = 256;
__device__ unsigned long long acx_pair_stamps[kPairStampBlocks * 2 * kPairStampSlots];
#define ACX_STAMP()                                                                                             \
    if (stamp_on && stamp_n < kPairStampSlots) {                                                                \
        unsigned long long t_;                                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                           \
        if (lane == 0 && stamp_n >= 0) acx_pair_stamps[(blockIdx.x * 2 + (wave >> 2)) * kPairStampSlots + stamp_n] = t_; \
        ++stamp_n;                                                                                              \
    }
#else
#define ACX_STAMP()
#endif

template <int C, int PT, bool LNOUT, bool ABF>
__global__ __launch_bounds__(512) void mlp_pair_bf16_kernel(
    const void* __restrict__ y, void* __restrict__ x, const char* __restrict__ wstream /*[2n][64 C bytes], pos order*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, int ld_out,
    __bf16* __restrict__ ln_out /* LNOUT: (M, ld_out) bf16 rows of LayerNorm(x_new), written INSTEAD of x */) {
    using Cfg = PairCfg<C, PT>;
    constexpr int n = Cfg::kChunks;
    constexpr int SEG = Cfg::kSegBytes;
    constexpr int P = Cfg::kSegPieces;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + Cfg::kOffB1);
    float* b2s = reinterpret_cast<float*>(smem + Cfg::kOffB2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    ACX_CLAIM_VGPR(255);          // CU-exclusive: 8 waves x 256 registers hold the SIMDs' whole register files
#ifdef ACX_PAIR_STAMPS
    const bool stamp_on = blockIdx.x < kPairStampBlocks && (wave & 3) == 0;
    int stamp_n = -3 * (4 * C / 32 + 1) * ACX_PAIR_STAMP_SKIP;      // the marks of the tile after the first ACX_PAIR_STAMP_SKIP tiles
#endif

    const unsigned smem_a = acx_lds_addr(smem);
    // The LDS-DMA pieces of the NEXT interval's two segments (stream positions pos0, pos1) are dealt over all eight waves --
    // piece i of wave w is q = w + 8 i of the 2 P -- and threaded through the MFMA loops of both roles: a piece holds its wave's
    // issue for 60-180 cycles, which the partner wave on the SIMD covers (as a burst at the top of an interval the requests
    // alone took a third of it: profiles/r05_b_pair_stamps.txt).  An interval that needs ONE segment passes it twice (3 of the
    // n + 1 intervals of a tile: the second copy re-requests the same bytes into the same place) -- one code path, and no
    // conditional around an MFMA loop (hipcc copies or spills every accumulator around such a branch).
    auto dma_piece = [&](const int i, const int pos0, const int pos1) __attribute__((always_inline)) {
        const int q = wave + 8 * i;
        const int second = q >= P ? 1 : 0;
        const int piece = q - second * P;
        const int pos = second ? pos1 : pos0;
        const char* src = wstream + (long long)pos * SEG + piece * 1024 + lane * 16;
#ifdef ACX_PAIR_VADDR        // (lab: the 64-bit per-lane address form, acx_glds16_own_m0 -- 15 % of the kernel: profiles/r05_e_lds_dma_saddr.txt)
        acx_glds16_own_m0(src, smem_a + (unsigned)((pos & 3) * SEG + piece * 1024));
#elif !defined(ACX_PAIR_NODMA)
        // scalar base + 32-bit lane offset (global_load_lds_dwordx4 v, s[..]): the piece's address is wave-uniform apart from 16 x lane
        (void)src;
        acx_glds16_s(wstream + (long long)pos * SEG + piece * 1024, lane * 16, smem_a + (unsigned)((pos & 3) * SEG + piece * 1024));
#else       // (NODMA: lab ablation, wrong results, timing only)
        asm volatile("" :: "v"(src));
#endif
    };
    constexpr int kCnt = 2 * P / 8;                 // pieces per wave and interval
    auto request = [&](const int pos0, const int pos1) __attribute__((always_inline)) {     // all of a wave's pieces at once
#pragma unroll
        for (int i = 0; i < kCnt; ++i) dma_piece(i, pos0, pos1);
    };
    static_assert(P % 4 == 0, "a segment pair's pieces are dealt over eight waves");
    // end of an interval: this wave's LDS writes and LDS-DMA pieces (and loads / stores) are done, then the workgroup meets
#define ACX_ENDINT { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

    for (int i = tid; i < 4 * C; i += 512) b1s[i] = 0.5f * b1[i];        // z = 0.5 v (gelu2h_micro)
    if (tid < C) b2s[tid] = b2[tid];
    request(0, 1);                                  // interval 0 of the first tile
    ACX_ENDINT

    const long long ntiles = (M + Cfg::kPix - 1) / Cfg::kPix;
    constexpr int kVar1 = 1 << (Cfg::kSwzBits - 1);
#ifndef ACX_PAIR_DEPTH
#define ACX_PAIR_DEPTH 3
#endif
    constexpr int kD = ACX_PAIR_DEPTH;               // rotating fragment registers: reads run kD - 1 units ahead of their MFMAs
#define ACX_B8(v_) __builtin_bit_cast(bf16x8p, v_)
#ifdef ACX_PAIR_NOMFMA      // (lab ablation)
#define ACX_MFMA(a_, b_, c_) ([&]() { asm volatile("" :: "v"(a_), "v"(b_)); return c_; }())
#else
#define ACX_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)
#endif
#ifdef ACX_PAIR_TRIPLE      // (lab: what would three MFMAs per fragment -- the fp32_split arithmetic -- make of this skeleton? timing only)
#define ACX_MFMA3(a_, b_, c_) ACX_MFMA(a_, b_, ACX_MFMA(a_, b_, ACX_MFMA(a_, b_, c_)))
#else
#define ACX_MFMA3(a_, b_, c_) ACX_MFMA(a_, b_, c_)
#endif

    if (wave < 4) {
        // ================================ producer ================================
        int w1off[kVar1];
#pragma unroll
        for (int q = 0; q < kVar1; ++q) w1off[q] = l31 * (2 * C) + (((2 * q + hh) ^ Cfg::swz1(l31)) << 4);
        GeluK3 gk = gelu_k2h();            // X holds z = 0.5 v itself: 0.5 W1 in the stream, 0.5 b1 staged
        gelu_k3_to_vgprs(gk);                       // two waves per SIMD: a scalar operand costs a vector instruction 2 extra cycles
        f32x4 act[PT][Cfg::kSteps];                 // lane (px = l31, half hh): channels 16 s + 8 hh .. + 7 as 8 bf16
        f32x16 Xa[PT], Xb[PT];                      // pre-activation tiles: one accumulates while the other's GELU is evaluated
        char* gbase = smem + Cfg::kOffG + pair * 2 * Cfg::kGBytes;

        // LayerNorm of the rows of a tile -> act (statistics in fp32: mean, then the centred sum of squares)
        auto load_ln = [&](const long long tile) __attribute__((always_inline)) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                long long r = tile * Cfg::kPix + pair * Cfg::kPairPix + pt * 32 + l31;
                if (r >= M) r = M - 1;
                if constexpr (ABF) {
                    const __bf16* yp = reinterpret_cast<const __bf16*>(y) + r * C + 8 * hh;
#pragma unroll
                    for (int s = 0; s < Cfg::kSteps; ++s) act[pt][s] = *reinterpret_cast<const f32x4*>(yp + 16 * s);
                    float sum = 0.f;
#pragma unroll
                    for (int s = 0; s < Cfg::kSteps; ++s) {
                        const uint4 u = __builtin_bit_cast(uint4, act[pt][s]);
                        sum += (acx_bf16_lo(u.x) + acx_bf16_hi(u.x)) + (acx_bf16_lo(u.y) + acx_bf16_hi(u.y));
                        sum += (acx_bf16_lo(u.z) + acx_bf16_hi(u.z)) + (acx_bf16_lo(u.w) + acx_bf16_hi(u.w));
                    }
                    sum += __shfl_xor(sum, 32);
                    const float mean = sum * (1.0f / C);
                    float d = 0.f;
#pragma unroll
                    for (int s = 0; s < Cfg::kSteps; ++s) {
                        const uint4 u = __builtin_bit_cast(uint4, act[pt][s]);
                        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const float t0 = acx_bf16_lo(w[p]) - mean, t1 = acx_bf16_hi(w[p]) - mean;
                            d = fmaf(t0, t0, d); d = fmaf(t1, t1, d);
                        }
                    }
                    d += __shfl_xor(d, 32);
                    const float rstd = 1.0f / sqrtf(d * (1.0f / C) + 1e-6f);
#pragma unroll
                    for (int s = 0; s < Cfg::kSteps; ++s) {
                        const uint4 u = __builtin_bit_cast(uint4, act[pt][s]);
                        const unsigned w[4] = {u.x, u.y, u.z, u.w};
                        unsigned o[4];
#pragma unroll
                        for (int p = 0; p < 4; ++p) o[p] = pair_pack_bf16((acx_bf16_lo(w[p]) - mean) * rstd, (acx_bf16_hi(w[p]) - mean) * rstd);
                        act[pt][s] = __builtin_bit_cast(f32x4, uint4{o[0], o[1], o[2], o[3]});
                    }
                } else {
                    const float* yp = reinterpret_cast<const float*>(y) + r * C + 8 * hh;
                    float a[C / 2];
#pragma unroll
                    for (int s = 0; s < Cfg::kSteps; ++s) {
                        const float4 v0 = *reinterpret_cast<const float4*>(yp + 16 * s), v1 = *reinterpret_cast<const float4*>(yp + 16 * s + 4);
                        a[8 * s] = v0.x; a[8 * s + 1] = v0.y; a[8 * s + 2] = v0.z; a[8 * s + 3] = v0.w;
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
                    const float rstd = 1.0f / sqrtf(d * (1.0f / C) + 1e-6f);
#pragma unroll
                    for (int s = 0; s < Cfg::kSteps; ++s) {
                        unsigned o[4];
#pragma unroll
                        for (int p = 0; p < 4; ++p) o[p] = pair_pack_bf16((a[8 * s + 2 * p] - mean) * rstd, (a[8 * s + 2 * p + 1] - mean) * rstd);
                        act[pt][s] = __builtin_bit_cast(f32x4, uint4{o[0], o[1], o[2], o[3]});
                    }
                }
            }
        };
        // G of a chunk: registers 8 sp .. 8 sp + 7 of a pixel tile's accumulator are this lane's 8 k values of phase 2's k-step sp
        unsigned un[PT][8];
        GeluState3 gst;
        // micro-step sg of the 64 PT that turn Xv into un: pixel tile sg / 64, register pair (sg % 64) / 8, step sg % 8 (7 of the GELU, 1 bf16 pack)
#define ACX_MICRO(Xv_, sg_)                                                                                     \
        {   constexpr int mt_ = (sg_) / 64, pr_ = ((sg_) % 64) / 8, st_ = (sg_) % 8;                            \
            const float ax_ = Xv_[mt_][2 * pr_], ay_ = Xv_[mt_][2 * pr_ + 1];                                   \
            if constexpr (st_ == 0) gelu2h_micro<0>(gst, gk, ax_, ay_);                                          \
            else if constexpr (st_ == 1) gelu2h_micro<1>(gst, gk, ax_, ay_);                                     \
            else if constexpr (st_ == 2) gelu2h_micro<2>(gst, gk, ax_, ay_);                                     \
            else if constexpr (st_ == 3) gelu2h_micro<3>(gst, gk, ax_, ay_);                                     \
            else if constexpr (st_ == 4) gelu2h_micro<4>(gst, gk, ax_, ay_);                                     \
            else if constexpr (st_ == 5) gelu2h_micro<5>(gst, gk, ax_, ay_);                                     \
            else if constexpr (st_ == 6) gelu2h_micro<6>(gst, gk, ax_, ay_);                                     \
            else un[mt_][pr_] = pair_pack_bf16(gst.qx, gst.qy); }
        auto write_g = [&](const int kc) __attribute__((always_inline)) {
            char* gs = gbase + (kc & 1) * Cfg::kGBytes + lane * 16;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
                    *reinterpret_cast<f32x4*>(gs + (pt * 2 + sp) * 1024) = __builtin_bit_cast(f32x4, uint4{un[pt][4 * sp], un[pt][4 * sp + 1], un[pt][4 * sp + 2], un[pt][4 * sp + 3]});
        };
        // One producer interval: Xn = b1 + W1c(kc) . act from ring slot `slot`; behind the MFMAs of every k-step ride this step's
        // share of the GELU of Xv (chunk kc - 1; HV: there is one) and of the wave's LDS-DMA pieces (next interval's segments rq0,
        // rq1, in the first two thirds of the loop: they must have landed when the interval ends); G(kc - 1) leaves at the end.
        auto phase1 = [&](auto hv_tag, f32x16 (&Xn)[PT], f32x16 (&Xv)[PT], const int kc, const int slot, const int rq0, const int rq1) __attribute__((always_inline)) {
            constexpr bool HV = decltype(hv_tag)::value;
            constexpr int kT = 64 * PT, kS = Cfg::kSteps, kSpan = (2 * kS) / 3;
            const char* base = smem + slot * SEG;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * kc + 8 * q + 4 * hh);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) { Xn[pt][4 * q] = bq[0]; Xn[pt][4 * q + 1] = bq[1]; Xn[pt][4 * q + 2] = bq[2]; Xn[pt][4 * q + 3] = bq[3]; }
            }
            // fragment reads run two k-steps ahead of their MFMAs in three rotating registers; the fences keep hipcc from hoisting
            // more of them (24 fragments in flight would be 96 registers) and the GELU steps in their gaps
#ifdef ACX_PAIR_NOREAD
#define ACX_W1_RD(s_) (act[0][(s_) % 4])
#else
#define ACX_W1_RD(s_) (*reinterpret_cast<const f32x4*>(base + ((s_) / kVar1) * (kVar1 * 32) + w1off[(s_) % kVar1]))
#endif
            f32x4 f[kD];
#pragma unroll
            for (int i = 0; i < kD - 1; ++i) f[i] = ACX_W1_RD(i);
            acx_static_for<kS>([&](auto s_tag) __attribute__((always_inline)) {
                constexpr int sx = decltype(s_tag)::value;
                if constexpr (sx + kD - 1 < kS) f[(sx + kD - 1) % kD] = ACX_W1_RD(sx + kD - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    Xn[pt] = ACX_MFMA3(ACX_B8(f[sx % kD]), ACX_B8(act[pt][sx]), Xn[pt]);
                if constexpr (sx < kSpan) {
#pragma unroll
                    for (int i = sx * kCnt / kSpan; i < (sx + 1) * kCnt / kSpan; ++i) dma_piece(i, rq0, rq1);
                }
#ifndef ACX_PAIR_NOGELU
                if constexpr (HV)
                    acx_static_for<kT * (sx + 1) / kS - kT * sx / kS>([&](auto g_tag) __attribute__((always_inline)) {
                        ACX_MICRO(Xv, kT * sx / kS + decltype(g_tag)::value)
                    });
#endif
                __builtin_amdgcn_sched_barrier(0);
            });
#undef ACX_W1_RD
            if constexpr (HV) write_g(kc - 1);
        };

        long long tile = blockIdx.x;
        if (tile < ntiles) load_ln(tile);
        for (; tile < ntiles; tile += gridDim.x) {
            const bool more = tile + gridDim.x < ntiles;
            // k = 0 (requests for interval 1: W1(1))
            ACX_STAMP() ACX_STAMP()
            phase1(std::false_type{}, Xa, Xb, 0, Cfg::pos_w1(0) & 3, Cfg::pos_w1(1), Cfg::pos_w1(1));
            ACX_STAMP()
            ACX_ENDINT
            // k = 1 (requests for interval 2: W1(2), W2(0))
            ACX_STAMP() ACX_STAMP()
            phase1(std::true_type{}, Xb, Xa, 1, Cfg::pos_w1(1) & 3, Cfg::pos_w1(2), Cfg::pos_w2(0));
            ACX_STAMP()
            ACX_ENDINT
            // k = 2 .. n - 1, two per trip: W1 in slot 3 (k even) / 1 (k odd); requests for interval k + 1: W1(k + 1), W2(k - 1)
            // (interval n reads W2(n - 2) only)
#pragma nounroll
            for (int k = 2; k < n; k += 2) {
                ACX_STAMP() ACX_STAMP()
                phase1(std::true_type{}, Xa, Xb, k, 3, 2 * k + 1, 2 * k + 2);
                ACX_STAMP()
                ACX_ENDINT
                const bool last = k + 1 == n - 1;
                ACX_STAMP() ACX_STAMP()
                phase1(std::true_type{}, Xb, Xa, k + 1, 1, last ? 2 * n - 1 : 2 * k + 3, last ? 2 * n - 1 : 2 * k + 4);
                ACX_STAMP()
                ACX_ENDINT
            }
            // k = n: the last GELU (chunk n - 1 sits in Xb: n - 1 is odd); requests for the next tile's interval 0; the next tile's rows
            ACX_STAMP()
            request(0, 1);
            acx_static_for<64 * PT>([&](auto g_tag) __attribute__((always_inline)) { ACX_MICRO(Xb, decltype(g_tag)::value) });
            write_g(n - 1);
            ACX_STAMP()
            if (more) load_ln(tile + gridDim.x);
            ACX_STAMP()
            ACX_ENDINT
        }
#undef ACX_MICRO
        // the consumer's two intervals behind the last tile
        ACX_ENDINT
        ACX_ENDINT
    } else {
        // ================================ consumer ================================
        int w2off[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) w2off[sp] = l31 * 64 + (((2 * sp + hh) ^ ((l31 >> 2) & 3)) << 4);
        f32x16 acc[PT][Cfg::kTiles];
        const char* gbase = smem + Cfg::kOffG + pair * 2 * Cfg::kGBytes + lane * 16;

        // out^T += W2c(chunk j) . G(j): the segment sits in ring slot `slot`, G in slot j & 1.  The pieces of the next interval's
        // segments (stream positions rq0, rq1; two_tag: one segment or two) go out between the MFMAs, in the first two thirds of
        // the loop (they must have landed when the interval ends)
        auto phase2 = [&](const int j, const int slot, const int rq0, const int rq1) __attribute__((always_inline)) {
            const char* base = smem + slot * SEG;
            constexpr int kUnits = 2 * Cfg::kTiles;
            constexpr int kSpan = (2 * kUnits) / 3;                                 // units that carry a piece
            f32x4 g[PT][2];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) g[pt][sp] = *reinterpret_cast<const f32x4*>(gbase + (j & 1) * Cfg::kGBytes + (pt * 2 + sp) * 1024);
#ifdef ACX_PAIR_NOREAD
#define ACX_W2_RD(u_) (g[0][(u_) & 1])
#else
#define ACX_W2_RD(u_) (*reinterpret_cast<const f32x4*>(base + ((u_) >> 1) * 2048 + w2off[(u_) & 1]))
#endif
            // unit u = (out tile t = u >> 1, k-step sp = u & 1); fragment reads two units ahead, fenced (see phase 1)
            f32x4 f[kD];
#pragma unroll
            for (int i = 0; i < kD - 1; ++i) f[i] = ACX_W2_RD(i);
#pragma unroll
            for (int u = 0; u < kUnits; ++u) {
                if (u + kD - 1 < kUnits) f[(u + kD - 1) % kD] = ACX_W2_RD(u + kD - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[pt][u >> 1] = ACX_MFMA3(ACX_B8(f[u % kD]), ACX_B8(g[pt][u & 1]), acc[pt][u >> 1]);
                // pieces [u kCnt / kSpan, (u + 1) kCnt / kSpan) ride behind unit u
#pragma unroll
                for (int i = (u < kSpan ? u * kCnt / kSpan : kCnt); i < (u + 1 < kSpan ? (u + 1) * kCnt / kSpan : kCnt) && u < kSpan; ++i)
                    dma_piece(i, rq0, rq1);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef ACX_W2_RD
        };
        // acc = x + b2 for the rows of `tile`.  bf16 rows (ABF): the residual enters through the matrix pipe -- acc = b2, then
        // acc += I . x with x read as phase-2-style B fragments (lane (px, hh): channels 32 t + 16 s' + 8 hh .. + 7, ONE 16-byte load)
        // and I the identity as an A fragment: 1.0 x bf16 is exact in the fp32 accumulate, 2 MFMAs per out tile (2 % of a tile's),
        // and 24 PT wide loads instead of 48 PT eight-byte ones in the accumulator layout, which cost 4 cycles per LANE of address
        // processing (profiles/r05_b_pair_stamps.txt: 47 k cycles per tile).  fp32 rows: direct loads in the accumulator layout.
        f32x4 ident[2];
        if constexpr (ABF) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const int e = l31 - 16 * sp - 8 * hh;         // this lane's row has its 1 at k = 16 s' + 8 hh + e
                unsigned w[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) w[p] = (e == 2 * p ? 0x3f80u : 0u) | (e == 2 * p + 1 ? 0x3f800000u : 0u);
                ident[sp] = __builtin_bit_cast(f32x4, uint4{w[0], w[1], w[2], w[3]});
            }
        }
        auto load_x = [&](const long long tile) __attribute__((always_inline)) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                long long r = tile * Cfg::kPix + pair * Cfg::kPairPix + pt * 32 + l31;
                if (r >= M) r = M - 1;
                if constexpr (ABF) {
                    const __bf16* xp = reinterpret_cast<const __bf16*>(x) + r * C + 8 * hh;
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 32 * t + 8 * q + 4 * hh);
                            acc[pt][t][4 * q + 0] = bb[0]; acc[pt][t][4 * q + 1] = bb[1]; acc[pt][t][4 * q + 2] = bb[2]; acc[pt][t][4 * q + 3] = bb[3];
                        }
#pragma unroll
                        for (int sp = 0; sp < 2; ++sp) {
                            const f32x4 xf = *reinterpret_cast<const f32x4*>(xp + 32 * t + 16 * sp);
                            acc[pt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ACX_B8(ident[sp]), ACX_B8(xf), acc[pt][t], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + r * C + 4 * hh + 32 * t + 8 * q);
                            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 32 * t + 8 * q + 4 * hh);
                            acc[pt][t][4 * q + 0] = v.x + bb[0]; acc[pt][t][4 * q + 1] = v.y + bb[1];
                            acc[pt][t][4 * q + 2] = v.z + bb[2]; acc[pt][t][4 * q + 3] = v.w + bb[3];
                        }
                }
            }
        };
        // acc (= x + b2 + out) -> x, or LayerNorm(acc) -> the downsample GEMM's bf16 operand rows
        auto epilogue = [&](const long long tile, auto masked_tag) __attribute__((always_inline)) {
            constexpr bool kMasked = decltype(masked_tag)::value;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const long long r = tile * Cfg::kPix + pair * Cfg::kPairPix + pt * 32 + l31;
                const bool valid = !kMasked || r < M;
                if constexpr (LNOUT) {
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int i = 0; i < 16; i += 4) sum += (acc[pt][t][i] + acc[pt][t][i + 1]) + (acc[pt][t][i + 2] + acc[pt][t][i + 3]);
                    sum += __shfl_xor(sum, 32);
                    const float mean = sum * (1.0f / C);
                    float d = 0.f;
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) { const float u = acc[pt][t][i] - mean; d = fmaf(u, u, d); }
                    d += __shfl_xor(d, 32);
                    const float rstd = 1.0f / sqrtf(d * (1.0f / C) + 1e-6f);
                    // lanes (px, 0) and (px, 1) trade pieces so that each writes 16 bytes (8 channels) per store
                    __bf16* op = ln_out + r * (long long)ld_out + 8 * hh;
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            unsigned e[2], o[2];
#pragma unroll
                            for (int w = 0; w < 2; ++w) {
                                e[w] = pair_pack_bf16((acc[pt][t][8 * j + 2 * w] - mean) * rstd, (acc[pt][t][8 * j + 2 * w + 1] - mean) * rstd);
                                o[w] = pair_pack_bf16((acc[pt][t][8 * j + 4 + 2 * w] - mean) * rstd, (acc[pt][t][8 * j + 4 + 2 * w + 1] - mean) * rstd);
                                acx_pair_swap(e[w], o[w]);
                            }
                            if (valid) *reinterpret_cast<uint4*>(op + 32 * t + 16 * j) = uint4{e[0], e[1], o[0], o[1]};
                        }
                    if (valid)
                        for (int c = C + 8 * hh; c < ld_out; c += 16)      // zero the K padding of the downsample GEMM's operand rows
                            *reinterpret_cast<uint4*>(ln_out + r * (long long)ld_out + c) = uint4{0u, 0u, 0u, 0u};
                } else {
#pragma unroll
                    for (int t = 0; t < Cfg::kTiles; ++t)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if constexpr (ABF) {        // bf16 rows: 16-byte pieces by trading halves with the partner lane
                                unsigned e[2] = {acx_pack_bf16x2(acc[pt][t][8 * j + 0], acc[pt][t][8 * j + 1]), acx_pack_bf16x2(acc[pt][t][8 * j + 2], acc[pt][t][8 * j + 3])};
                                unsigned o[2] = {acx_pack_bf16x2(acc[pt][t][8 * j + 4], acc[pt][t][8 * j + 5]), acx_pack_bf16x2(acc[pt][t][8 * j + 6], acc[pt][t][8 * j + 7])};
                                acx_pair_swap(e[0], o[0]);
                                acx_pair_swap(e[1], o[1]);
                                if (valid)
                                    *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(x) + r * C + 8 * hh + 32 * t + 16 * j) = uint4{e[0], e[1], o[0], o[1]};
                            } else if (valid) {
                                float* xp = reinterpret_cast<float*>(x) + r * C + 4 * hh + 32 * t + 16 * j;
                                *reinterpret_cast<float4*>(xp) = make_float4(acc[pt][t][8 * j + 0], acc[pt][t][8 * j + 1], acc[pt][t][8 * j + 2], acc[pt][t][8 * j + 3]);
                                *reinterpret_cast<float4*>(xp + 8) = make_float4(acc[pt][t][8 * j + 4], acc[pt][t][8 * j + 5], acc[pt][t][8 * j + 6], acc[pt][t][8 * j + 7]);
                            }
                        }
                }
            }
        };

        // the epilogue's stores stay in flight across the barrier of their interval (counted wait: they are the youngest operations
        // of the wave, behind its LDS-DMA pieces) -- for a tile whose rows all exist and that stores x (a fixed number of instructions)
        constexpr int kStores = (ABF ? 2 : 4) * Cfg::kTiles * PT;
#define ACX_ENDINT_KEEP(n_) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(n_) : "memory"); __builtin_amdgcn_sched_barrier(0); }
        auto leave = [&](const long long t) __attribute__((always_inline)) {     // the tile's results leave; ends the interval
            if (!LNOUT && (t + 1) * Cfg::kPix <= M) {
                epilogue(t, std::false_type{});
                ACX_ENDINT_KEEP(kStores < 64 ? kStores : 63)
            } else {
                epilogue(t, std::true_type{});
                ACX_ENDINT
            }
        };
        long long prev = -1;
        for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            // k = 0: the previous tile's last chunk (first tile: the same instructions on whatever the registers hold -- the result
            // is overwritten by load_x below; no branch around an MFMA loop), then its results leave; requests for interval 1: W1(1)
            ACX_STAMP() ACX_STAMP()
            phase2(n - 1, Cfg::pos_w2(n - 1) & 3, Cfg::pos_w1(1), Cfg::pos_w1(1));
            ACX_STAMP()
            if (prev >= 0) leave(prev);
            else ACX_ENDINT
            // k = 1: this tile's residual rows arrive; requests for interval 2: W1(2), W2(0)
            ACX_STAMP()
            request(Cfg::pos_w1(2), Cfg::pos_w2(0));
            ACX_STAMP()
            load_x(tile);
            ACX_STAMP()
            ACX_ENDINT
            // k = 2 .. n - 1: W2(k - 2) sits in slot 0 (k even) / 2 (k odd); requests for interval k + 1: W1(k + 1), W2(k - 1)
            // (interval n reads W2(n - 2) only)
#pragma nounroll
            for (int k = 2; k < n; k += 2) {
                ACX_STAMP() ACX_STAMP()
                phase2(k - 2, 0, 2 * k + 1, 2 * k + 2);
                ACX_STAMP()
                ACX_ENDINT
                const bool last = k + 1 == n - 1;
                ACX_STAMP() ACX_STAMP()
                phase2(k - 1, 2, last ? 2 * n - 1 : 2 * k + 3, last ? 2 * n - 1 : 2 * k + 4);
                ACX_STAMP()
                ACX_ENDINT
            }
            // k = n; requests for the next tile's interval 0: W1(0), W2(n - 1)
            ACX_STAMP() ACX_STAMP()
            phase2(n - 2, Cfg::pos_w2(n - 2) & 3, 0, 1);
            ACX_STAMP()
            ACX_ENDINT
            prev = tile;
        }
        phase2(n - 1, Cfg::pos_w2(n - 1) & 3, 0, 1);      // (its requests are idle: nothing reads them)
        leave(prev);
        ACX_ENDINT
#undef ACX_ENDINT_KEEP
    }
#undef ACX_B8
#undef ACX_MFMA
#undef ACX_MFMA3
#undef ACX_ENDINT
}

// number of CUs of the current device (persistent launches), cached per device
static int pair_cu_count() {
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

template <int C, int PT, bool LNOUT, bool ABF>
static int launch_pair_cfg(const BlockW& w, const void* y, void* x, long long M, void* ln_out, int ld_out, hipStream_t s) {
    using Cfg = PairCfg<C, PT>;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "ring + G + biases do not fit the LDS");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_pair_bf16_kernel<C, PT, LNOUT, ABF>, kCuLdsBytes));
    const long long tiles = (M + Cfg::kPix - 1) / Cfg::kPix;
    // one persistent workgroup per CU of this launch's share of the chip (sub-batches run side by side on two streams)
    long long share = pair_cu_count() / inflight_ways();
    if (share < 1) share = 1;
    const long long blocks = tiles < share ? tiles : share;
    launch_kernel(&mlp_pair_bf16_kernel<C, PT, LNOUT, ABF>, dim3((unsigned)blocks), dim3(512), kCuLdsBytes /* CU-exclusive */, s,
        y, x, reinterpret_cast<const char*>(w.wstream_b), w.b1, w.b2, M, ld_out, reinterpret_cast<__bf16*>(ln_out));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

bool mlp_pair_bf16_supported(int C) { return C == 384 || C == 192; }
int mlp_pair_bf16_swz(int C, int row) { return C == 384 ? PairCfg<384, 1>::swz1(row) : PairCfg<192, 2>::swz1(row); }
int mlp_pair_bf16_pos_w1(int C, int k) { return C == 384 ? PairCfg<384, 1>::pos_w1(k) : PairCfg<192, 2>::pos_w1(k); }
int mlp_pair_bf16_pos_w2(int C, int j) { return C == 384 ? PairCfg<384, 1>::pos_w2(j) : PairCfg<192, 2>::pos_w2(j); }

template <bool ABF>
static int launch_pair_any(const BlockW& w, int C, const void* y, void* x, long long M, hipStream_t s, void* ln_out, int ld_out) {
    if (C == 384) return ln_out ? launch_pair_cfg<384, 1, true, ABF>(w, y, x, M, ln_out, ld_out, s) : launch_pair_cfg<384, 1, false, ABF>(w, y, x, M, nullptr, 0, s);
    if (C == 192) return ln_out ? launch_pair_cfg<192, 2, true, ABF>(w, y, x, M, ln_out, ld_out, s) : launch_pair_cfg<192, 2, false, ABF>(w, y, x, M, nullptr, 0, s);
    ACX_FAIL(ACX_ERR_SHAPE, "paired bf16 MLP: unsupported channel count %d", C);
}

int launch_mlp_pair_bf16(acx_ctx* c, const BlockW& w, int C, const void* y, void* x, long long M, hipStream_t s,
                         void* ln_out, int ld_out, bool act_bf16) {
    if (!w.wstream_b) ACX_FAIL(ACX_ERR_STATE, "paired bf16 MLP: the weight stream was not packed for C=%d", C);
    ProfScope ps(c, ACX_K_MLP_WIDE, s);
    return act_bf16 ? launch_pair_any<true>(w, C, y, x, M, s, ln_out, ld_out) : launch_pair_any<false>(w, C, y, x, M, s, ln_out, ld_out);
}

}  // namespace acx
