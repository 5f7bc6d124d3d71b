// EXPERIMENTAL, NOT BUILT INTO libacx (round 2): correct (parity suite green for C = 96) but SLOWER than the kernels
// that ship -- 678 us per stage-0 block against 595 for the 8-wave mlp_fused_split_kernel<96>.  The lab
// (tools/run_widep_lab.sh) shows why: with every activation load removed it still takes 588 us; at C = 96 a segment is 18
// MFMAs (576 cycles) but carries 36 GELU micro-steps (~650 cycles of vector issue on the one wave of the SIMD), three
// LDS-DMA issues (~100 cycles each) and a barrier + its bookkeeping (~250): the prefetch works (I/O costs only 90 us of
// 678), the loop is vector-issue- and overhead-bound.  Kept for the record and as the starting point for a
// two-pixel-tile variant.
//
// K4fp -- the wide fused block MLP (mlp_fused_wide.hip: same arithmetic, same weight stream, same three-stream
// schedule) as a PERSISTENT kernel with tile prefetch, for the narrow stages (C = 96, 192) where a block is close to
// HBM-bound: per 128-pixel tile the matrix work is 12 (C = 96) or 24 (C = 192) chunks, about as long as reading y and
// x and writing x takes, and a CU-exclusive workgroup has no neighbour on its CU to overlap the two.  So one workgroup
// per CU walks over its tiles (blockIdx.x, + gridDim.x, ...) and, while tile t multiplies,
//   * its residual x(t) and the next tile's y(t + gridDim.x) are loaded into registers that C <= 192 leaves free
//     (C/2 each of the 512 per lane);
//   * the weight stream keeps running across the tile boundary through a DEEPER ring (R slots, a segment is requested
//     R - 1 segments ahead): one wave's vector-memory counter is in issue order, so a wait for a weight piece also
//     waits for every load issued before it -- the prefetch loads therefore get R - 1 segments (a few microseconds) to
//     land, not one.
// The prefetch loads are inline asm: hipcc does not know they are in flight, so it neither drains the queue
// (s_waitcnt vmcnt(0)) where their registers are first used -- a tile later, long after the segment-end waits have
// covered them -- nor miscounts: untracked operations only ever make the compiler's own counted waits stricter.
// Segment-end waits count exactly: the weight pieces requested since, plus the prefetch loads and epilogue stores
// issued since (their positions in a tile are fixed and every one of them is issued unconditionally).
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {

template <int C, int R>
struct WidePCfg {
    static constexpr int kWaves = 4;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kPix = kWaves * 32;
    static constexpr int kChunks = 4 * C / 32;              // n
    static constexpr int kSegs = 2 * kChunks;
    static constexpr int kSegBytes = 128 * C;               // one [32][C] or [C][32] S16 image
    static constexpr int kPieces = kSegBytes / 1024 / kWaves;   // 1-KB LDS-DMA pieces per wave per segment
    static constexpr int kSteps = C / 16;                   // units of a phase-1 segment (k-steps)
    static constexpr int kUnits = 2 * (C / 32);             // units of a phase-2 segment (out tile, k-step)
    static constexpr int kMfmas = 3 * kUnits;               // MFMAs per segment
    static constexpr int kLook = R - 1;                     // segments a request runs ahead of its use
    static constexpr size_t kLdsBytes = R * (size_t)kSegBytes + 4 * C * 4;
    static_assert(C % 32 == 0 && kSteps == kUnits && kUnits % kPieces == 0 && (36 % kMfmas == 0 || kMfmas % 36 == 0), "unit / piece bookkeeping");
    static_assert(kPieces * kLook <= 63, "the segment-end wait must be encodable in s_waitcnt vmcnt");
    // W1 rows are 4 C bytes: the XOR that spreads 16 consecutive rows over the banks (see mlp_fused_wide.hip / api.hip)
    __device__ static int swz1(int row) { return (C % 64 == 0) ? (row & 15) : ((row >> 1) & 7); }
};

template <int C, bool LNOUT, int R>
__global__ __launch_bounds__(256) void mlp_fused_widep_kernel(
    const float* __restrict__ y, float* __restrict__ x, const char* __restrict__ wstream /*[2n][128 C bytes]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, int ntiles, float sinv1, float sinv2,
    float hscale, char* __restrict__ ln_out /* LNOUT: (M, C) S16 rows of LayerNorm(x_new) x 2^11, written INSTEAD of x */) {
    using Cfg = WidePCfg<C, R>;
    constexpr int L = Cfg::kLook;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + R * Cfg::kSegBytes);   // [4C], pre-divided by sinv1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    ACX_CLAIM_VGPR(255);          // CU-exclusive: one wave per SIMD holds the SIMD's whole register file
    ACX_CLAIM_AGPR(255);
    constexpr int n = Cfg::kChunks;
    // this workgroup's tiles: blockIdx.x, + gridDim.x, ...; its weight segments are numbered g = 0 .. total - 1 across tiles
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * Cfg::kSegs;

    const int dma_lane = (wave * Cfg::kPieces) * 1024 + lane * 16;      // this lane's slot in piece 0 of its wave
#define ACX_WDMA(sseg_, piece_, slot_)                                                                           \
        __builtin_amdgcn_global_load_lds(                                                                        \
            (const __attribute__((address_space(1))) void*)(wstream + (long long)(sseg_) * Cfg::kSegBytes + dma_lane + (piece_) * 1024), \
            (__attribute__((address_space(3))) void*)(smem + (slot_) * Cfg::kSegBytes + (wave * Cfg::kPieces + (piece_)) * 1024), 16, 0, 0);
    // segments 0 .. L-1 are requested before anything else; segment g + L follows during segment g
#pragma unroll
    for (int q = 0; q < L; ++q)
        if (q < total) {
#pragma unroll
            for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(q % Cfg::kSegs, p, q % R)
        }
    {
        const float b1scale = 1.0f / sinv1;             // a power of two
        for (int i = tid; i < 4 * C; i += Cfg::kThreads) b1s[i] = b1[i] * b1scale;
    }

    // fragment addresses inside a segment (without the ring offset), as in mlp_fused_wide.hip
    const int sw1 = Cfg::swz1(l31);
    int w1off[4][2], w2off[2][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w1off[q][pl] = l31 * (4 * C) + (((4 * q + 2 * hh + pl) ^ sw1) << 4);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w2off[sp][pl] = l31 * 128 + (((2 * (2 * sp + hh) + pl) ^ ((l31 >> 1) & 7)) << 4);
    GeluConsts gk;
    gk.ps = 0.3275911f * 0.70710678f * sinv1;
    gk.cq = 0.84932180f * sinv1;       // sqrt(log2(e) / 2): exp(-v^2 / 2) = exp2(-(cq a)^2)
    gk.ca = -0.5f * sinv1 * hscale;
    gk.cb = sinv1 * hscale;

#define ACX_H8(v_) __builtin_bit_cast(h8, v_)
#define ACX_FENCE __builtin_amdgcn_sched_barrier(0);
    // phase-1 unit = k-step s_ of the chunk: chunk 4 s_ + ..: the bits above the XORed four = s_ / 4 -> + 256 B each
#define ACX_W1_RD(base_, s_, pl_) (*reinterpret_cast<const f32x4*>((base_) + ((s_) >> 2) * 256 + w1off[(s_) & 3][pl_]))
    // MFMA number m_ of a segment is followed (behind a scheduling fence) by its share of the 36 GELU micro-steps the
    // segment carries: steps [36 m / kMfmas, 36 (m + 1) / kMfmas) of the half (C = 384: one after every other MFMA,
    // C = 192: one after each)
#define ACX_AFTER_MFMA(HV_, half_, m_)                                                                          \
        ACX_FENCE if constexpr (HV_) { ACX_MICRO_RANGE(36 * (half_) + 36 * (m_) / Cfg::kMfmas, 36 * (half_) + 36 * ((m_) + 1) / Cfg::kMfmas) } ACX_FENCE
#define ACX_P1_MFMA(s_, ah_, al_)                                                                               \
        Xn = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(acth[s_]), Xn, 0, 0, 0);                \
        ACX_AFTER_MFMA(HV, 1, 3 * (s_) + 0)                                                                     \
        Xn = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(actl[s_]), Xn, 0, 0, 0);                \
        ACX_AFTER_MFMA(HV, 1, 3 * (s_) + 1)                                                                     \
        Xn = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(acth[s_]), Xn, 0, 0, 0);                \
        ACX_AFTER_MFMA(HV, 1, 3 * (s_) + 2)
    // phase-2 unit i = (out tile t = i >> 1, k-step s' = i & 1)
#define ACX_W2_RD(base_, i_, pl_) (*reinterpret_cast<const f32x4*>((base_) + ((i_) >> 1) * 4096 + w2off[(i_) & 1][pl_]))
#define ACX_P2_MFMA(i_, ah_, al_)                                                                               \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(gh[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER_MFMA(HV, 0, 3 * (i_) + 0)                                                                     \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(gl[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER_MFMA(HV, 0, 3 * (i_) + 1)                                                                     \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(gh[(i_) & 1]), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER_MFMA(HV, 0, 3 * (i_) + 2)
    // micro-steps [from, to) of the 72 (8 register pairs x 9 steps) that turn Xv into uh / ul
#define ACX_MICRO_RANGE(from_, to_)                                                                             \
        _Pragma("unroll") for (int mm_ = (from_); mm_ < (to_); ++mm_) {                                         \
            const int pr_ = mm_ / 9, st_ = mm_ - 9 * pr_;                                                       \
            if (st_ == 0) { gs.ax = Xv[2 * pr_]; gs.ay = Xv[2 * pr_ + 1]; gelu_micro<0>(gs, gk, uh[pr_], ul[pr_]); } \
            else if (st_ == 1) gelu_micro<1>(gs, gk, uh[pr_], ul[pr_]);                                         \
            else if (st_ == 2) gelu_micro<2>(gs, gk, uh[pr_], ul[pr_]);                                         \
            else if (st_ == 3) gelu_micro<3>(gs, gk, uh[pr_], ul[pr_]);                                         \
            else if (st_ == 4) gelu_micro<4>(gs, gk, uh[pr_], ul[pr_]);                                         \
            else if (st_ == 5) gelu_micro<5>(gs, gk, uh[pr_], ul[pr_]);                                         \
            else if (st_ == 6) gelu_micro<6>(gs, gk, uh[pr_], ul[pr_]);                                         \
            else if (st_ == 7) gelu_micro<7>(gs, gk, uh[pr_], ul[pr_]);                                         \
            else gelu_micro<8>(gs, gk, uh[pr_], ul[pr_]);                                                       \
        }
#define ACX_TOUCH2(h_, l_) { asm volatile("" :: "v"(h_)); asm volatile("" :: "v"(l_)); }
#define ACX_BIAS_INIT(j_)                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * (j_) + 8 * q + 4 * hh);                 \
            Xn[4 * q + 0] = bq[0]; Xn[4 * q + 1] = bq[1]; Xn[4 * q + 2] = bq[2]; Xn[4 * q + 3] = bq[3];         \
        }
#define ACX_PACK_G()                                                                                            \
        gh[0] = __builtin_bit_cast(f32x4, uint4{uh[0], uh[1], uh[2], uh[3]});                                   \
        gh[1] = __builtin_bit_cast(f32x4, uint4{uh[4], uh[5], uh[6], uh[7]});                                   \
        gl[0] = __builtin_bit_cast(f32x4, uint4{ul[0], ul[1], ul[2], ul[3]});                                   \
        gl[1] = __builtin_bit_cast(f32x4, uint4{ul[4], ul[5], ul[6], ul[7]});
    constexpr int kLoads = C / 8;                        // float4 loads per lane for one tile's rows (y or x)
    constexpr int kStores = LNOUT ? C / 4 : C / 8;       // store instructions of the epilogue
    // end of segment g: every wave's pieces of segment g + 1 must have landed and every wave must be done reading segment
    // g before its slot is requested again.  The pieces of g + 1 were requested during segment g + 1 - L; whatever was
    // issued after them may stay in flight: rem_ segments' pieces (L - 1 in steady state) and extra_ prefetch / store
    // instructions.  Only the steady-state counts are special-cased; anything else waits for more than it needs.
    auto seg_end = [&](const int rem_, const int extra_) __attribute__((always_inline)) {
        ACX_FENCE
        constexpr int W = Cfg::kPieces * (L - 1);
#define ACX_WAIT_X(e_) else if (rem_ >= L - 1 && extra_ == (e_) && W + (e_) <= 63) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W + (e_) <= 63 ? W + (e_) : 0) : "memory");
#define ACX_WAIT_CASE(r_) else if (L - 1 > (r_) && rem_ == (r_)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::kPieces * ((r_) < L ? (r_) : 0)) : "memory");
        if (rem_ >= L - 1 && extra_ == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W) : "memory");
        ACX_WAIT_X(kLoads) ACX_WAIT_X(2 * kLoads) ACX_WAIT_X(kStores) ACX_WAIT_X(kLoads + kStores) ACX_WAIT_X(2 * kLoads + kStores)
        else if (rem_ >= L - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W) : "memory");
        ACX_WAIT_CASE(1) ACX_WAIT_CASE(2) ACX_WAIT_CASE(3) ACX_WAIT_CASE(4) ACX_WAIT_CASE(5) ACX_WAIT_CASE(6)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ACX_WAIT_CASE
#undef ACX_WAIT_X
        __builtin_amdgcn_s_barrier();
        ACX_FENCE
    };
    // Positions inside a tile (segment numbers 0 .. kSegs - 1) after which the extra vector-memory instructions are issued:
    // the residual loads after segment 0, the next tile's y loads after segment kYPos, the epilogue stores after the last.
    constexpr int kXPos = 0, kYPos = Cfg::kSegs / 2 - 1, kSPos = Cfg::kSegs - 1;
    int pos = 0;                  // position of the running segment inside its tile
    bool first_tile = true, has_next = false;
    // extra instructions issued after the pieces of segment g + 1 (requested during position pos + 1 - L, possibly in
    // the previous tile) and before the end of segment pos: those issued after positions e in [pos + 1 - L, pos - 1]
    auto extra_now = [&]() -> int {
        int e = 0;
        const int lo = pos + 1 - L, hi = pos - 1;
        auto in = [&](int q) { return q >= lo && q <= hi; };
        if (in(kXPos)) e += kLoads;                                   // this tile's residual loads
        if (in(kYPos) && has_next) e += kLoads;                       // the next tile's y loads, issued in this tile
        if (!first_tile) {                                            // the previous tile: its y loads (always issued there) and stores
            if (in(kYPos - Cfg::kSegs)) e += kLoads;
            if (in(kSPos - Cfg::kSegs)) e += kStores;
        }
        return e;
    };

    f32x16 Xn, Xv;        // Xn: pre-activation being accumulated by phase 1; Xv: the previous chunk's, input of the GELU
    f32x4 gh[2], gl[2];   // G(k - 1): B operand of phase 2, two k-steps, hi / lo halves
    unsigned uh[8], ul[8];
    constexpr int kDmaStride = Cfg::kUnits / Cfg::kPieces;       // one piece every kDmaStride units
    GeluState gs;
    f32x4 acth[Cfg::kSteps], actl[Cfg::kSteps];         // this wave's normalised activations, 8 fp16 halves each
    f32x16 acc[C / 32];

    int g = 0;            // this workgroup's running segment number
    int slot = 0;         // g % R
    int sseg = 0;         // g % kSegs: position in the weight stream
    // one phase-1 segment: Xn = b1 + W1c . LN(y)^T for chunk k_; requests segment g + L
    auto phase1 = [&](auto with_gelu, const int k_) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;     // second half of the GELU of Xv rides on this segment's MFMAs
        const char* base = smem + slot * Cfg::kSegBytes;
        const bool dma = g + L < total;
        const int rq_slot = slot == 0 ? R - 1 : slot - 1;                 // (g + L) % R
        int rq_seg = sseg + L; rq_seg = rq_seg >= Cfg::kSegs ? rq_seg - Cfg::kSegs : rq_seg;      // L < kSegs
        ACX_BIAS_INIT(k_)
        f32x4 a0h = ACX_W1_RD(base, 0, 0), a0l = ACX_W1_RD(base, 0, 1), a1h, a1l;
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; s += 2) {
            a1h = ACX_W1_RD(base, s + 1, 0); a1l = ACX_W1_RD(base, s + 1, 1);
            ACX_FENCE
            ACX_P1_MFMA(s, a0h, a0l)
            if (s % kDmaStride == 0 && dma) { ACX_WDMA(rq_seg, s / kDmaStride, rq_slot) }
            ACX_FENCE
            ACX_TOUCH2(a1h, a1l)
            if (s + 2 < Cfg::kSteps) { a0h = ACX_W1_RD(base, s + 2, 0); a0l = ACX_W1_RD(base, s + 2, 1); }
            ACX_FENCE
            ACX_P1_MFMA(s + 1, a1h, a1l)
            if ((s + 1) % kDmaStride == 0 && dma) { ACX_WDMA(rq_seg, (s + 1) / kDmaStride, rq_slot) }
            ACX_FENCE
            if (s + 2 < Cfg::kSteps) ACX_TOUCH2(a0h, a0l)
        }
        if constexpr (HV) { ACX_PACK_G() }
        Xv = Xn;
        seg_end(total - 2 - g, extra_now());
        ++g; ++pos; slot = slot == R - 1 ? 0 : slot + 1; sseg = sseg == Cfg::kSegs - 1 ? 0 : sseg + 1;
    };
    // one phase-2 segment: out^T += W2c . G for the chunk whose G sits in gh / gl; with_gelu: the first half of the GELU +
    // split of Xv (the NEXT chunk) rides on this segment's MFMAs
    auto phase2 = [&](auto with_gelu) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;
        const char* base = smem + slot * Cfg::kSegBytes;
        const bool dma = g + L < total;
        const int rq_slot = slot == 0 ? R - 1 : slot - 1;
        int rq_seg = sseg + L; rq_seg = rq_seg >= Cfg::kSegs ? rq_seg - Cfg::kSegs : rq_seg;
        f32x4 a0h = ACX_W2_RD(base, 0, 0), a0l = ACX_W2_RD(base, 0, 1), a1h, a1l;
#pragma unroll
        for (int i = 0; i < Cfg::kUnits; i += 2) {
            a1h = ACX_W2_RD(base, i + 1, 0); a1l = ACX_W2_RD(base, i + 1, 1);
            ACX_FENCE
            ACX_P2_MFMA(i, a0h, a0l)
            if (i % kDmaStride == 0 && dma) { ACX_WDMA(rq_seg, i / kDmaStride, rq_slot) }
            ACX_FENCE
            ACX_TOUCH2(a1h, a1l)
            if (i + 2 < Cfg::kUnits) { a0h = ACX_W2_RD(base, i + 2, 0); a0l = ACX_W2_RD(base, i + 2, 1); }
            ACX_FENCE
            ACX_P2_MFMA(i + 1, a1h, a1l)
            if ((i + 1) % kDmaStride == 0 && dma) { ACX_WDMA(rq_seg, (i + 1) / kDmaStride, rq_slot) }
            ACX_FENCE
            if (i + 2 < Cfg::kUnits) ACX_TOUCH2(a0h, a0l)
        }
        seg_end(total - 2 - g, extra_now());
        ++g; ++pos; slot = slot == R - 1 ? 0 : slot + 1; sseg = sseg == Cfg::kSegs - 1 ? 0 : sseg + 1;
    };

    // lane (px = l31, half hh) of a tile: y channels 16s + 8hh .. +7 (s = 0 .. C/16 - 1) feed phase 1; x channels
    // 32t + 8q + 4hh .. +3 meet the accumulators in the epilogue
    f32x4 yn[C / 8];              // raw y rows of the NEXT tile (of the first tile before the loop): [2 s + half]
    f32x4 xr[C / 8];              // residual x of the CURRENT tile: [4 t + q]
    // asynchronous 16-byte load the compiler does not track (see the header); offset_ is a compile-time byte offset
#define ACX_ALOAD(dst_, ptr_, offset_) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(dst_) : "v"(ptr_), "n"(offset_) : "memory");
    auto load_y = [&](const int tile_) __attribute__((always_inline)) {
        long long row = (long long)tile_ * Cfg::kPix + wave * 32 + l31;
        if (row >= M) row = M - 1;
        const float* yp = y + row * C + 8 * hh;
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) {
            ACX_ALOAD(yn[2 * s], yp, 64 * s)
            ACX_ALOAD(yn[2 * s + 1], yp, 64 * s + 16)
        }
    };
    load_y((int)blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first tile's rows: nothing to overlap them with
    __syncthreads();      // segments 0 .. L-1 landed; b1s visible

    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        long long mrow = (long long)tile * Cfg::kPix + wave * 32 + l31;
        if (mrow >= M) mrow = M - 1;
        // ---- LayerNorm of this tile's y rows (in yn) -> acth / actl --------------------------------------------------------
        {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < C / 2; ++i) sum += yn[i >> 2][i & 3];
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / C);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < C / 2; ++i) { const float t = yn[i >> 2][i & 3] - mean; d = fmaf(t, t, d); }
            d += __shfl_xor(d, 32);
            const float sc = kSplitLnScale / sqrtf(d * (1.0f / C) + 1e-6f);
#pragma unroll
            for (int s = 0; s < Cfg::kSteps; ++s) {
                unsigned uh4[4], ul4[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x2 v;
                    v.x = (yn[2 * s + (p >> 1)][2 * (p & 1)] - mean) * sc; v.y = (yn[2 * s + (p >> 1)][2 * (p & 1) + 1] - mean) * sc;
                    const h2 h = __builtin_convertvector(v, h2);
                    const f32x2 back = __builtin_convertvector(h, f32x2);
                    const h2 l = __builtin_convertvector(v - back, h2);
                    uh4[p] = __builtin_bit_cast(unsigned, h);
                    ul4[p] = __builtin_bit_cast(unsigned, l);
                }
                acth[s] = __builtin_bit_cast(f32x4, uint4{uh4[0], uh4[1], uh4[2], uh4[3]});
                actl[s] = __builtin_bit_cast(f32x4, uint4{ul4[0], ul4[1], ul4[2], ul4[3]});
            }
        }
#pragma unroll
        for (int t = 0; t < C / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        // segment 0 of the tile: phase 1 of chunk 0; the first half of its GELU has nothing to ride on
        phase1(std::false_type{}, 0);
        ACX_MICRO_RANGE(0, 36)
        // prefetch (the loads get L segments to land, see the header): this tile's residual, the next tile's y rows
        {
            const float* xp = x + mrow * C + 4 * hh;
#pragma unroll
            for (int t = 0; t < C / 32; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) { ACX_ALOAD(xr[4 * t + q], xp, (32 * t + 8 * q) * 4) }
        }
        const int next_tile = tile + (int)gridDim.x;
        has_next = next_tile < ntiles;
        for (int k = 1; k < n; ++k) {
            phase1(std::true_type{}, k);                    // position 2k - 1
            if (2 * k - 1 == kYPos && has_next) load_y(next_tile);
            phase2(std::true_type{});
        }
        ACX_MICRO_RANGE(36, 72)         // second half of the last chunk's GELU: no phase-1 segment left to ride on
        ACX_PACK_G()
        phase2(std::false_type{});

        // ---- epilogue: lane (px, hh), tile t, q: channels 32t + 8q + 4hh .. +3  ->  x = x + out + b2 -------------------
        if constexpr (LNOUT) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < C / 32; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * t + 8 * q;
                    const float4 bb = *reinterpret_cast<const float4*>(b2 + c + 4 * hh);
                    const f32x4 v = xr[4 * t + q];
                    acc[t][4 * q + 0] = v[0] + fmaf(acc[t][4 * q + 0], sinv2, bb.x);
                    acc[t][4 * q + 1] = v[1] + fmaf(acc[t][4 * q + 1], sinv2, bb.y);
                    acc[t][4 * q + 2] = v[2] + fmaf(acc[t][4 * q + 2], sinv2, bb.z);
                    acc[t][4 * q + 3] = v[3] + fmaf(acc[t][4 * q + 3], sinv2, bb.w);
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
            {   // every lane stores: a lane past the end holds row M - 1 (clamped) and writes the same bytes as its owner
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
        } else {    // every lane stores (clamped rows write identical values): the store count per tile is fixed
            float* xp = x + mrow * C + 4 * hh;
#pragma unroll
            for (int t = 0; t < C / 32; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * t + 8 * q;
                    const float4 bb = *reinterpret_cast<const float4*>(b2 + c + 4 * hh);
                    f32x4 v = xr[4 * t + q];
                    v[0] += fmaf(acc[t][4 * q + 0], sinv2, bb.x);
                    v[1] += fmaf(acc[t][4 * q + 1], sinv2, bb.y);
                    v[2] += fmaf(acc[t][4 * q + 2], sinv2, bb.z);
                    v[3] += fmaf(acc[t][4 * q + 3], sinv2, bb.w);
                    *reinterpret_cast<f32x4*>(xp + c) = v;
                }
            }
        }
        pos = 0;
        first_tile = false;
    }
#undef ACX_ALOAD
#undef ACX_WDMA
#undef ACX_H8
#undef ACX_FENCE
#undef ACX_W1_RD
#undef ACX_P1_MFMA
#undef ACX_W2_RD
#undef ACX_P2_MFMA
#undef ACX_TOUCH2
#undef ACX_BIAS_INIT
#undef ACX_MICRO_RANGE
#undef ACX_AFTER_MFMA
#undef ACX_PACK_G
}

template <int C, bool LNOUT, int R>
static int launch_widep_cfg(const BlockW& w, const float* y, float* x, long long M, void* ln_out, hipStream_t s) {
    using Cfg = WidePCfg<C, R>;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "weight ring does not fit the LDS");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_widep_kernel<C, LNOUT, R>, kCuLdsBytes));
    static int n_cu[64] = {0};          // compute units per device (persistent grid = one workgroup per CU)
    int dev = 0;
    ACX_HIP(hipGetDevice(&dev));
    if (n_cu[dev & 63] == 0) {
        hipDeviceProp_t prop;
        ACX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu[dev & 63] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long ntiles = (M + Cfg::kPix - 1) / Cfg::kPix;
    if (ntiles > 0x7fffffffLL) ACX_FAIL(ACX_ERR_SHAPE, "fused MLP: too many tiles");
    const long long blocks = ntiles < n_cu[dev & 63] ? ntiles : n_cu[dev & 63];
    mlp_fused_widep_kernel<C, LNOUT, R><<<dim3((unsigned)blocks), dim3(Cfg::kThreads), kCuLdsBytes /* CU-exclusive */, s>>>(
        y, x, reinterpret_cast<const char*>(w.wstream_s), w.b1, w.b2, M, (int)ntiles, 1.0f / (kSplitLnScale * w.w1s_scale),
        1.0f / (w.hid_scale * w.w2s_scale), w.hid_scale, reinterpret_cast<char*>(ln_out));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// C = 192 is NOT served here although it builds: with 4 x C/2 resident registers the allocator starts moving the
// asynchronously loaded prefetch registers (VGPR <-> AGPR copies right behind the load instruction, i.e. before the data
// has landed) and the block returns NaN; with compiler-tracked loads instead it drains the queue at every tile and is
// slower than the non-persistent kernel (522 vs 469 us).  mlp_fused_wide.hip serves C = 192.
bool mlp_fused_widep_supported(int C) { return C == 96; }

int launch_mlp_fused_widep(acx_ctx* c, const BlockW& w, int C, const float* y, float* x, long long M, hipStream_t s,
                           void* ln_out) {
    if (!w.wstream_s) ACX_FAIL(ACX_ERR_STATE, "persistent fused MLP: the weight stream was not packed for C=%d", C);
    ProfScope ps(c, ACX_K_MLP_FUSED, s);
    // ring depth: as many 128 C-byte segments as the LDS holds next to the bias table, at most 8
    if (C == 96) return ln_out ? launch_widep_cfg<96, true, 8>(w, y, x, M, ln_out, s) : launch_widep_cfg<96, false, 8>(w, y, x, M, nullptr, s);
    if (C == 192) return ln_out ? launch_widep_cfg<192, true, 6>(w, y, x, M, ln_out, s) : launch_widep_cfg<192, false, 6>(w, y, x, M, nullptr, s);
    ACX_FAIL(ACX_ERR_SHAPE, "persistent fused MLP: unsupported channel count %d", C);
}

}  // namespace acx
