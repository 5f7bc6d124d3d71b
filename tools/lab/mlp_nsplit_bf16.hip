// LAB KERNEL (round 6; builds only inside tools/lab/ns_lab.hip, not part of libacx.so): parity-green through the block emulation
// tests while it was wired into run_block (git history: "N-split fused bf16 MLP"), 160-163 us per C = 384 block at B = 64 against
// 162-163 for the ring kernel, 3-5 % faster at 16 x the pixels, 1.3 % SLOWER in the two-stream forward -- so it does not ship.
// What it showed is in DESIGN.md 3i and profiles/r06_a_nsplit_lab.txt.
//
// K4n -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86) of the `bf16a`
// arithmetic (bf16 MFMA operands AND bf16 activations in HBM, BASELINE configs[2]) for the wide stages, round 6.
//
// What bounded the ring kernel of mlp_fused_wide_bf16.hip: with one 32-pixel tile per wave every weight fragment
// read from the LDS (1 KB) feeds ONE v_mfma_f32_32x32x16_bf16 (32 cycles), all four waves read every segment, and 4 KB of LDS
// reads per 32 matrix cycles is the LDS's whole bandwidth; and a tile's rows entered and left the registers ROW PER LANE in the
// MFMA operand layouts -- 32 lines per vector-memory instruction, ~4 cycles of address processing each, 288 such instructions
// per 128-pixel tile: the 38 k cycles of prologue + epilogue of a 175 k-cycle tile.  Two pixel tiles per wave (every fragment
// feeds two MFMAs) need 576 registers at C = 384.  This kernel gets the reuse by splitting N instead:
//   * the four waves of a CU-exclusive workgroup form two PAIRS; a pair owns 64 pixels (two tiles of 32), BOTH of its waves hold
//     the normalised activations of all 64 (C/2 registers);
//   * phase 1 (X = W1c . LN(y)^T, chunk of 64 hidden units): wave r of the pair computes hidden units 32 r .. 32 r + 31 for both
//     pixel tiles -- one W1 fragment per k-step, two MFMAs;
//   * GELU in registers, G = bf16(GELU(X)) of the wave's 32 hidden units goes to a 16-KB LDS exchange buffer in the lane order in
//     which it is phase 2's B operand (4 KB per wave and chunk); phase 2 reads all 64 hidden units' G back (own + partner's);
//   * phase 2 (out^T += W2c . G): wave r accumulates output channels C/2 r .. C/2 r + C/2 - 1 of both pixel tiles (C/2
//     registers instead of C) -- one W2 fragment per (out tile, k-step), two MFMAs.
//   Per chunk and wave: the same 96 MFMAs (C = 384) as before, 48 fragment reads instead of 96, + 12 KB of G traffic.
//   * tile I/O through the LDS, 1 KB per instruction: the tile's y rows arrive by LDS-DMA (global_load_lds_dwordx4, lanes on
//     consecutive 16-byte pieces of whole rows) into the two ring slots that are idle at the start, each wave normalises 32 rows
//     out of the LDS and hands them to its partner through the same bytes; the residual rows arrive the same way under the last
//     two segments and enter the accumulators through the matrix pipe (out += I . x: exact, 1.0 x bf16 in the fp32 accumulate);
//     the result leaves through the LDS as whole rows.
// Weight stream, segment order, LDS images and rounding points are those of mlp_fused_wide_bf16.hip (api.hip packs ONE
// `wstream_b` for both kernels): LayerNorm output and GELU output rounded to bf16 (round to nearest even), weights once at
// acx_finalize, fp32 statistics / accumulation / GELU / residual add; the emulation tests of tests/test_gpu_bf16.py run on it.
// The pwconv1 bias enters as the C operand of a chunk's first MFMAs (scalar loads of 0.5 b1 -- the stream holds 0.5 W1, the
// accumulator is z = 0.5 v, split_math.h gelu2h), the pwconv2 bias as the initial value of the out accumulators: no bias in the
// LDS, which is full (3 x 128 C ring + 16 KB = 160 KB at C = 384).
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {
bool mlp_nsplit_bf16_supported(int C, bool ln_out);
// w.b1 must hold 0.5 b1 (the stream holds 0.5 W1: the accumulator is z = 0.5 v, split_math.h gelu2h)
int launch_mlp_nsplit_bf16(acx_ctx* c, const BlockW& w, int C, const void* y, void* x, long long M, hipStream_t s);
}

namespace acx {

typedef __bf16 nbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 nbf16x2 __attribute__((ext_vector_type(2)));

template <int C>
struct NsCfg {
    static constexpr int kWaves = 4;
    static constexpr int kThreads = 256;
    static constexpr int kPix = 128;                        // pixels of a workgroup tile: 2 pairs x 2 tiles of 32
    static constexpr int kChunks = 4 * C / 64;              // n: chunks of 64 hidden units
    static constexpr int kSegs = 2 * kChunks;
    static constexpr int kSegBytes = 128 * C;               // [64][C] or [C][64] bf16
    static constexpr int kPieces = kSegBytes / 1024 / kWaves;
    static constexpr int kSteps = C / 16;                   // phase 1: k-steps = fragment reads of a segment (2 MFMAs each)
    static constexpr int kTW = C / 64;                      // phase 2: out tiles of 32 channels per wave
    static constexpr int kUnits = kSteps;                   // phase 2: (k-step 0..3, out tile) = 4 kTW = kSteps fragment reads
    static constexpr int kMfmas = 2 * kUnits;               // per segment and wave
    static constexpr int kDmaStride = kUnits / kPieces;
    static constexpr int kRowBytes = 2 * C;                 // a pixel row of bf16 activations
    static constexpr int kRowChunks = C / 8;                // ... in 16-byte chunks
    static constexpr int kTileBytes = kPix * kRowBytes;     // = 2 kSegBytes: a tile's rows fill two ring slots
    static constexpr int kTilePieces = kTileBytes / 1024 / kWaves;
    static constexpr int kStageY = kSegBytes;               // y rows: slots 1-2 (slot 0 receives segment 0 meanwhile)
    static constexpr int kStageX = 0;                       // x rows / result rows: slots 0-1 (the last segment sits in slot 2)
    static constexpr int kOffG = 3 * kSegBytes;             // G exchange: [pair][tile half][k-step 0..3][64 lanes x 16 B]
    static constexpr int kGBytes = 16 * 1024;
    static constexpr size_t kLdsBytes = (size_t)kOffG + kGBytes;
    static constexpr int kGeluSteps = 8 * 7;                // per 32 x 32 tile: 8 register pairs x 7 micro-steps
    static_assert(kSegs % 3 == 0, "the last segment must sit in ring slot 2");
    static_assert(kUnits % kPieces == 0 && kTileBytes == 2 * kSegBytes, "piece bookkeeping");
    // rows of 2 C bytes: the XOR that spreads 16 consecutive rows over the LDS banks (as WideBfCfg: the W1 images are shared)
    static constexpr int kSwzBits = (C % 128 == 0) ? 4 : ((C % 64 == 0) ? 3 : 2);
    __host__ __device__ static int swz1(int row) { return kSwzBits == 4 ? (row & 15) : (kSwzBits == 3 ? ((row >> 1) & 7) : ((row >> 2) & 3)); }
};

__device__ __forceinline__ unsigned ns_pack_bf16(float a, float b) {
    f32x2 v; v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, nbf16x2));
}

// gelu2h (split_math.h) cut into six steps of two instructions: the register pair (ax, ay) holds z = 0.5 v, the result is left in
// (qx, qy); a seventh step packs it to bf16
template <int ST>
__device__ __forceinline__ void ns_gelu_step(float& qx, float& qy, const GeluK3 k, const float ax, const float ay) {
    if constexpr (ST == 0) { qx = __builtin_fmaf(__builtin_fabsf(ax), k.k2, k.k1); qy = __builtin_fmaf(__builtin_fabsf(ay), k.k2, k.k1); }
    else if constexpr (ST == 1) { qx = __builtin_fmaf(qx, __builtin_fabsf(ax), k.k0); qy = __builtin_fmaf(qy, __builtin_fabsf(ay), k.k0); }
    else if constexpr (ST == 2) { qx *= __builtin_fabsf(ax); qy *= __builtin_fabsf(ay); }
    else if constexpr (ST == 3) { qx = __builtin_amdgcn_exp2f(qx); qy = __builtin_amdgcn_exp2f(qy); }
    else if constexpr (ST == 4) { qx = 1.0f - qx; qy = 1.0f - qy; }
    else { qx = __builtin_fmaf(__builtin_fabsf(ax), qx, ax); qy = __builtin_fmaf(__builtin_fabsf(ay), qy, ay); }
}

#ifndef ACX_NS_NOGELU
#define ACX_NS_NOGELU 0     // lab ablations (tools/lab/ns_lab.hip): wrong results, timing only
#endif
#ifndef ACX_NS_NOMFMA
#define ACX_NS_NOMFMA 0
#endif
#ifdef ACX_NS_STAMPS        // lab builds only: s_memtime at the marks of wave 0 of the first kNsStampBlocks workgroups
constexpr int kNsStampBlocks = 64, kNsStampSlots = 400;
__device__ unsigned long long acx_ns_stamps[kNsStampBlocks * kNsStampSlots];
#define ACX_NS_STAMP()                                                                                          \
    if (stamp_on && stamp_n < kNsStampSlots) {                                                                  \
        unsigned long long t_;                                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                           \
        if (lane == 0) acx_ns_stamps[blockIdx.x * kNsStampSlots + stamp_n] = t_;                                \
        ++stamp_n;                                                                                              \
    }
#else
#define ACX_NS_STAMP()
#endif
#if defined(ACX_NS_STAMPS) && defined(ACX_NS_STAMPS_FINE)
#define ACX_NS_FINE() ACX_NS_STAMP()
#else
#define ACX_NS_FINE()
#endif

template <int C, bool LNOUT>
__global__ __launch_bounds__(256) void mlp_nsplit_bf16_kernel(
    const __bf16* __restrict__ y, __bf16* __restrict__ x, const char* __restrict__ wstream /*[2n][128 C bytes]*/,
    const float* __restrict__ b1h /*[4C]: 0.5 b1*/, const float* __restrict__ b2, long long M, int ld_out,
    __bf16* __restrict__ ln_out /* LNOUT: (M, ld_out) bf16 rows of LayerNorm(x_new), written INSTEAD of x */) {
    using Cfg = NsCfg<C>;
    constexpr int n = Cfg::kChunks;
    constexpr int RB = Cfg::kRowBytes;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int ph = wave >> 1, r = wave & 1;       // pair, role inside the pair
    ACX_CLAIM_VGPR(255);          // CU-exclusive: one wave per SIMD holds the SIMD's whole register file
    ACX_CLAIM_AGPR(255);
    const long long row0 = (long long)blockIdx.x * Cfg::kPix;
    const int rows_valid = (int)((M - row0) < (long long)Cfg::kPix ? (M - row0) : (long long)Cfg::kPix);

#ifdef ACX_NS_STAMPS
    const bool stamp_on = blockIdx.x < kNsStampBlocks && wave == 0;
    int stamp_n = 0;
#endif
    ACX_NS_STAMP()      // 0: start
    const unsigned smem_a = acx_lds_addr(smem);
    const unsigned dma_voff = lane * 16;
    // LDS-DMA pieces with a scalar base and a 32-bit lane offset, issued from inline asm: the counted waits are this file's own
#define ACX_WDMA(seg_, piece_, slot_)                                                                            \
        acx_glds16_s(wstream + (long long)(seg_) * Cfg::kSegBytes + (wave * Cfg::kPieces + (piece_)) * 1024, dma_voff,   \
                     smem_a + (unsigned)((slot_) * Cfg::kSegBytes + (wave * Cfg::kPieces + (piece_)) * 1024));
    // A tile's rows in the LDS: row-major, kRowChunks chunks of 16 B per row, chunk c of row w at position c ^ swz1(w) (the W1
    // image's geometry: the fragment reads below are that image's).  Lane offset of LDS chunk `gidx` (linear) in the tile's
    // rows in HBM; rows past M repeat the last valid row (finite values, never stored).
    auto stage_voff = [&](const int gidx) __attribute__((always_inline)) -> unsigned {
        const int row = gidx / Cfg::kRowChunks, pos = gidx - row * Cfg::kRowChunks;
        const int c = pos ^ Cfg::swz1(row);
        const int rc = row < rows_valid ? row : rows_valid - 1;
        return (unsigned)(rc * RB + c * 16);
    };
    const char* ytile = acx_scalar_ptr(reinterpret_cast<const char*>(y) + row0 * RB);
    const char* xtile = acx_scalar_ptr(reinterpret_cast<const char*>(x) + row0 * RB);

#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(0, p, 0)
#pragma unroll
    for (int q = 0; q < Cfg::kTilePieces; ++q) {
        const int piece = wave * Cfg::kTilePieces + q;
        acx_glds16_s(ytile, stage_voff(piece * 64 + lane), smem_a + (unsigned)(Cfg::kStageY + piece * 1024));
    }
#define ACX_LDS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ACX_LDS_BARRIER
    ACX_NS_STAMP()      // 1: y rows + segment 0 landed

    // ---- LayerNorm: this wave normalises the pair's rows 32 r .. 32 r + 31 (its tile 0), writes them back as bf16 and reads the
    // partner's (its tile 1).  Lane (px = l31, half hh) holds channels 16 s + 8 hh .. + 7 of k-step s: the B operand of phase 1.
    const int rown = 64 * ph + 32 * r + l31, rowp = 64 * ph + 32 * (1 - r) + l31;        // tile-local rows
    f32x4 act[2][Cfg::kSteps];
    {
        char* rp = smem + Cfg::kStageY + rown * RB;
        const int sx = Cfg::swz1(rown);
        float a[C / 2];
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) {
            const uint4 u = *reinterpret_cast<const uint4*>(rp + (((2 * s + hh) ^ sx) << 4));
            a[8 * s + 0] = acx_bf16_lo(u.x); a[8 * s + 1] = acx_bf16_hi(u.x); a[8 * s + 2] = acx_bf16_lo(u.y); a[8 * s + 3] = acx_bf16_hi(u.y);
            a[8 * s + 4] = acx_bf16_lo(u.z); a[8 * s + 5] = acx_bf16_hi(u.z); a[8 * s + 6] = acx_bf16_lo(u.w); a[8 * s + 7] = acx_bf16_hi(u.w);
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
            for (int p = 0; p < 4; ++p) u4[p] = ns_pack_bf16((a[8 * s + 2 * p] - mean) * rstd, (a[8 * s + 2 * p + 1] - mean) * rstd);
            act[0][s] = __builtin_bit_cast(f32x4, uint4{u4[0], u4[1], u4[2], u4[3]});
            *reinterpret_cast<f32x4*>(rp + (((2 * s + hh) ^ sx) << 4)) = act[0][s];
        }
    }
    ACX_LDS_BARRIER
    ACX_NS_STAMP()      // 2: own rows normalised, written back
    {
        const char* rp = smem + Cfg::kStageY + rowp * RB;
        const int sx = Cfg::swz1(rowp);
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; ++s) act[1][s] = *reinterpret_cast<const f32x4*>(rp + (((2 * s + hh) ^ sx) << 4));
    }
    ACX_LDS_BARRIER           // the staged rows are dead: slots 1-2 belong to the ring from here on
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(1, p, 1)

    // ---- out accumulators of this wave's C/2 channels, both tiles: start from the pwconv2 bias (scalar loads) ----
    f32x16 acc[2][Cfg::kTW];
#pragma unroll
    for (int t = 0; t < Cfg::kTW; ++t) {
        const float* bp = b2 + (C / 2) * r + 32 * t;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = bp[8 * q + e], hi = bp[8 * q + 4 + e];
                const float v = hh ? hi : lo;
                acc[0][t][4 * q + e] = v; acc[1][t][4 * q + e] = v;
            }
    }

    // fragment addresses inside a segment (see mlp_fused_wide_bf16.hip for the images):
    //   W1: row = hidden unit 32 r + l31 of the chunk, k-step s = chunk 2 s + hh at position ^ swz1(l31)
    //   W2: row = out channel (128 B), this wave's tile t = rows 32 (kTW r + t) + l31, k-step s' = chunk 2 s' + hh at ^ ((l31 >> 1) & 7)
    const int w1row = (32 * r + l31) * RB;
    const int w1x = (hh << 4) ^ (Cfg::swz1(l31) << 4);
    int w2off[4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) w2off[sp] = (Cfg::kTW * r) * 4096 + l31 * 128 + (((2 * sp + hh) ^ ((l31 >> 1) & 7)) << 4);
    // G exchange: [pair][absolute tile half][k-step][lane]; this wave's tile i is the pair's half r ^ i
    char* gown[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) gown[i] = smem + Cfg::kOffG + ph * 8192 + (r ^ i) * 4096 + lane * 16;
    const GeluK3 gk = gelu_k2h();
    ACX_NS_STAMP()      // 3: partner's rows read, accumulators initialised

#define ACX_B8(v_) __builtin_bit_cast(nbf16x8, v_)
#define ACX_FENCE __builtin_amdgcn_sched_barrier(0);
#if defined(ACX_NS_NOREAD)
    const f32x4 fake_f = act[0][0];
#define ACX_W1_RD(base_, u_) fake_f
#define ACX_W2_RD(base_, u_) fake_f
#else
#define ACX_W1_RD(base_, u_) (*reinterpret_cast<const f32x4*>((base_) + ((32 * (u_)) ^ w1x)))
#define ACX_W2_RD(base_, u_) (*reinterpret_cast<const f32x4*>((base_) + ((u_) % Cfg::kTW) * 4096 + w2off[(u_) / Cfg::kTW]))
#endif
#define ACX_G_RD(i_, sp_) (*reinterpret_cast<const f32x4*>(gown[i_] + (sp_) * 1024))
    // GELU micro-steps [from, to) of ONE tile's kGeluSteps (8 register pairs x 7 steps): two pairs in flight, their steps
    // alternating (consecutive steps never depend on each other); register pair p of X_ -> un[tile_][p]
#if ACX_NS_NOGELU
#define ACX_MICRO_RANGE(X_, tile_, from_, to_)                                                                  \
        _Pragma("unroll") for (int sg_ = (from_); sg_ < (to_); ++sg_) {                                         \
            const int gp_ = sg_ / 14, w_ = sg_ % 14, st_ = w_ >> 1, pr_ = 2 * gp_ + (w_ & 1);                   \
            if (st_ == 6) un[tile_][pr_] = ns_pack_bf16(X_[2 * pr_], X_[2 * pr_ + 1]); }
#else
#define ACX_MICRO_RANGE(X_, tile_, from_, to_)                                                                  \
        _Pragma("unroll") for (int sg_ = (from_); sg_ < (to_); ++sg_) {                                         \
            const int gp_ = sg_ / 14, w_ = sg_ % 14, st_ = w_ >> 1, wh_ = w_ & 1, pr_ = 2 * gp_ + wh_;          \
            const float ax_ = X_[2 * pr_], ay_ = X_[2 * pr_ + 1];                                               \
            if (st_ == 0) ns_gelu_step<0>(gq[wh_][0], gq[wh_][1], gk, ax_, ay_);                                \
            else if (st_ == 1) ns_gelu_step<1>(gq[wh_][0], gq[wh_][1], gk, ax_, ay_);                           \
            else if (st_ == 2) ns_gelu_step<2>(gq[wh_][0], gq[wh_][1], gk, ax_, ay_);                           \
            else if (st_ == 3) ns_gelu_step<3>(gq[wh_][0], gq[wh_][1], gk, ax_, ay_);                           \
            else if (st_ == 4) ns_gelu_step<4>(gq[wh_][0], gq[wh_][1], gk, ax_, ay_);                           \
            else if (st_ == 5) ns_gelu_step<5>(gq[wh_][0], gq[wh_][1], gk, ax_, ay_);                           \
            else un[tile_][pr_] = ns_pack_bf16(gq[wh_][0], gq[wh_][1]);                                         \
        }
#endif
    // the share of MFMA number m_ (0 .. kMfmas - 1) of a segment
    // (the first kGeluHead steps of a segment's share run BEFORE its first MFMA, in the shadow of the segment's first LDS reads --
    // the latency nothing else covers behind a segment boundary)
#ifndef ACX_NS_GELU_HEAD
#define ACX_NS_GELU_HEAD 8
#endif
    constexpr int kGeluHead = ACX_NS_GELU_HEAD;
#define ACX_MICRO_AFTER(X_, tile_, m_) ACX_MICRO_RANGE(X_, tile_, kGeluHead + (Cfg::kGeluSteps - kGeluHead) * (m_) / Cfg::kMfmas, kGeluHead + (Cfg::kGeluSteps - kGeluHead) * ((m_) + 1) / Cfg::kMfmas)
#define ACX_MICRO_HEAD(X_, tile_) ACX_MICRO_RANGE(X_, tile_, 0, kGeluHead)
#if ACX_NS_NOMFMA
#define ACX_MFMA(a_, b_, c_) (c_); asm volatile("" :: "v"(a_), "v"(b_));
#else
#define ACX_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(ACX_B8(a_), ACX_B8(b_), c_, 0, 0, 0);
#endif
#define ACX_TOUCH1(f_) { asm volatile("" :: "v"(f_)); }
    // Weight pieces in RUNS (split_math.h): the wave's kPieces pieces of a segment are 1 KB apart in the stream and in the LDS,
    // so one M0 write and one scalar base serve up to eight of them and a piece is ONE instruction.  Nothing else touches M0
    // while a run is open (the residual-row pieces of the last two segments set it per piece, and no run is open there).
#if defined(ACX_NS_NODMA)
#define ACX_WDMA_LOOP(seg_, p_, slot_)
#else
#define ACX_WDMA_LOOP(seg_, p_, slot_) {                                                                        \
        const int run0_ = ((p_) / 8) * 8;                                                                       \
        if ((p_) == run0_) acx_set_m0(smem_a + (unsigned)((slot_) * Cfg::kSegBytes + (wave * Cfg::kPieces + run0_ + 4) * 1024)); \
        acx_glds16_run_s(wstream + (long long)(seg_) * Cfg::kSegBytes + (wave * Cfg::kPieces + run0_ + 4) * 1024, dma_voff, (p_) - run0_); }
#endif
    // end of a segment: every piece but this segment's own kPieces has landed (in-order completion) -- i.e. the NEXT segment --,
    // this wave's LDS writes (G) are done, then the workgroup meets
#if defined(ACX_NS_NODMA) || defined(ACX_NS_NOWAIT)
#define ACX_SEG_WAIT_N 63
#else
#define ACX_SEG_WAIT_N Cfg::kPieces
#endif
#if defined(ACX_NS_NOBAR)
#define ACX_SEG_BAR ""
#else
#define ACX_SEG_BAR "\n\ts_barrier"
#endif
#define ACX_SEG_END(last_)                                                                                      \
        ACX_FENCE                                                                                               \
        if (last_) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                     \
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ACX_SEG_BAR :: "n"(ACX_SEG_WAIT_N) : "memory");       \
        ACX_FENCE
#define ACX_WRITE_G()                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                        \
        _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_)                                                        \
            *reinterpret_cast<f32x4*>(gown[i_] + (2 * r + m_) * 1024) =                                         \
                __builtin_bit_cast(f32x4, uint4{un[i_][4 * m_ + 0], un[i_][4 * m_ + 1], un[i_][4 * m_ + 2], un[i_][4 * m_ + 3]});

    f32x16 Xn[2];                 // pre-activation tiles of the chunk in phase 1: hidden units 32 r .., this wave's two pixel tiles
    f32x16 Xv1;                   // tile 1 of the chunk before, while its GELU rides on phase 1 (tile 0's rides on phase 2, out of Xn[0])
    unsigned un[2][8];            // G of the chunk under construction: this wave's k-steps 2 r, 2 r + 1 of phase 2
    float gq[2][2];
    // The pwconv1 bias of a chunk (x 0.5): C operand of the chunk's first MFMAs.  Lane (px, hh) holds hidden units 8 q + 4 hh + e
    // of the tile in register 4 q + e: 32 scalar loads and a select per register.  Loaded a whole segment ahead and selected
    // behind the MFMAs of the phase 2 in front (exposed at the top of phase 1 it was 800 cycles of a 3 300-cycle chunk).
    f32x16 biasv;
    float sb[32];
#if defined(ACX_NS_NOBIAS)
#define ACX_SB_LOAD(k_)
#define ACX_SB_TOUCH()
#define ACX_BIAS_SEL(j_) biasv[j_] = 0.f;
#else
#define ACX_SB_LOAD(k_) { const float* bp_ = b1h + 64 * (k_) + 32 * r; _Pragma("unroll") for (int i_ = 0; i_ < 32; ++i_) sb[i_] = bp_[i_]; }
    // (the loads are issued where ACX_SB_LOAD stands only if their results are needed soon after: this empty asm is that use)
#define ACX_SB_TOUCH()                                                                                          \
        asm volatile("" : "+s"(sb[0]), "+s"(sb[1]), "+s"(sb[2]), "+s"(sb[3]), "+s"(sb[4]), "+s"(sb[5]), "+s"(sb[6]), "+s"(sb[7]),       \
                          "+s"(sb[8]), "+s"(sb[9]), "+s"(sb[10]), "+s"(sb[11]), "+s"(sb[12]), "+s"(sb[13]), "+s"(sb[14]), "+s"(sb[15])); \
        asm volatile("" : "+s"(sb[16]), "+s"(sb[17]), "+s"(sb[18]), "+s"(sb[19]), "+s"(sb[20]), "+s"(sb[21]), "+s"(sb[22]), "+s"(sb[23]), \
                          "+s"(sb[24]), "+s"(sb[25]), "+s"(sb[26]), "+s"(sb[27]), "+s"(sb[28]), "+s"(sb[29]), "+s"(sb[30]), "+s"(sb[31]));
#define ACX_BIAS_SEL(j_) biasv[j_] = hh ? sb[8 * ((j_) >> 2) + 4 + ((j_) & 3)] : sb[8 * ((j_) >> 2) + ((j_) & 3)];
#endif

    // phase 1 of chunk k_ (X = W1c . act, C operand of the first MFMAs = biasv) in ring slot slot_.  HV: the GELU of Xv1 (tile 1
    // of the chunk before) rides on these MFMAs, and that chunk's finished G goes to the exchange buffer at the end.
    auto phase1 = [&](auto with_gelu, const int seg_, const int slot_) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;
        const char* base = smem + slot_ * Cfg::kSegBytes + w1row;
        const int slot2 = slot_ == 0 ? 2 : slot_ - 1;        // (slot_ + 2) % 3
        f32x4 f[3];
        f[0] = ACX_W1_RD(base, 0);
        f[1] = ACX_W1_RD(base, 1);
        ACX_FENCE
        if constexpr (HV) { ACX_MICRO_HEAD(Xv1, 1) }
        ACX_NS_FINE()
#pragma unroll
        for (int u = 0; u < Cfg::kUnits; ++u) {
            if (u == 8 || u == 16) { ACX_NS_FINE() }
            if (u + 2 < Cfg::kUnits) f[(u + 2) % 3] = ACX_W1_RD(base, u + 2);
            ACX_FENCE
            if (u == 0) { Xn[0] = ACX_MFMA(f[0], act[0][0], biasv) } else { Xn[0] = ACX_MFMA(f[u % 3], act[0][u], Xn[0]) }
            ACX_FENCE
            if constexpr (HV) { ACX_MICRO_AFTER(Xv1, 1, 2 * u) }
            ACX_FENCE
            if (u == 0) { Xn[1] = ACX_MFMA(f[0], act[1][0], biasv) } else { Xn[1] = ACX_MFMA(f[u % 3], act[1][u], Xn[1]) }
            ACX_FENCE
            if constexpr (HV) { ACX_MICRO_AFTER(Xv1, 1, 2 * u + 1) }
            if (u % Cfg::kDmaStride == 0) { ACX_WDMA_LOOP(seg_ + 2, u / Cfg::kDmaStride, slot2) }
            ACX_FENCE
            if (u + 1 < Cfg::kUnits) ACX_TOUCH1(f[(u + 1) % 3])
        }
        if constexpr (HV) { ACX_WRITE_G() }
        ACX_NS_FINE()
        ACX_SEG_END(false)
    };
    // phase 2 (out^T += W2c . G) of the chunk whose G sits in the exchange buffer, ring slot slot_.  HV: the GELU of Xn[0] (tile 0
    // of the NEXT chunk) rides on these MFMAs and Xn[1] moves to Xv1 at the end.  NB: the bias of chunk kb_ (the next phase 1) is
    // loaded and selected here.  XD = 0: this segment requests the pieces of segment seg_ + 2; 1 / 2: the first / second half of
    // the tile's residual rows instead (the last two segments of a tile)
    auto phase2 = [&](auto with_gelu, auto next_bias, auto xdma, const int kb_, const int seg_, const int slot_) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;
        constexpr bool NB = decltype(next_bias)::value;
        constexpr int XD = decltype(xdma)::value;
        const char* base = smem + slot_ * Cfg::kSegBytes;
        const int slot2 = slot_ == 0 ? 2 : slot_ - 1;
        if constexpr (NB) { ACX_SB_LOAD(kb_) }
        f32x4 gb[2][2];
        gb[0][0] = ACX_G_RD(0, 0); gb[0][1] = ACX_G_RD(1, 0);
        f32x4 f[3];
        f[0] = ACX_W2_RD(base, 0);
        f[1] = ACX_W2_RD(base, 1);
        ACX_FENCE
        if constexpr (HV) { ACX_MICRO_HEAD(Xn[0], 0) }
        ACX_NS_FINE()
#pragma unroll
        for (int u = 0; u < Cfg::kUnits; ++u) {
            const int sp = u / Cfg::kTW, t = u % Cfg::kTW;
            if (u == 8 || u == 16) { ACX_NS_FINE() }
            if (t == 0 && sp + 1 < 4) { gb[(sp + 1) & 1][0] = ACX_G_RD(0, sp + 1); gb[(sp + 1) & 1][1] = ACX_G_RD(1, sp + 1); }
            if (u + 2 < Cfg::kUnits) f[(u + 2) % 3] = ACX_W2_RD(base, u + 2);
            ACX_FENCE
            acc[0][t] = ACX_MFMA(f[u % 3], gb[sp & 1][0], acc[0][t])
            ACX_FENCE
            if constexpr (HV) { ACX_MICRO_AFTER(Xn[0], 0, 2 * u) }
            if constexpr (NB) { if (u >= Cfg::kUnits - 8) { ACX_BIAS_SEL(2 * (u - (Cfg::kUnits - 8))) } }
            ACX_FENCE
            acc[1][t] = ACX_MFMA(f[u % 3], gb[sp & 1][1], acc[1][t])
            ACX_FENCE
            if constexpr (HV) { ACX_MICRO_AFTER(Xn[0], 0, 2 * u + 1) }
            if constexpr (NB) { if (u == Cfg::kUnits / 3) { ACX_SB_TOUCH() } if (u >= Cfg::kUnits - 8) { ACX_BIAS_SEL(2 * (u - (Cfg::kUnits - 8)) + 1) } }
            if (u % Cfg::kDmaStride == 0) {
                if constexpr (XD == 0) { ACX_WDMA_LOOP(seg_ + 2, u / Cfg::kDmaStride, slot2) }
                else {
                    const int piece = (XD - 1) * (Cfg::kTilePieces * 2) + wave * Cfg::kPieces + u / Cfg::kDmaStride;
                    acx_glds16_s(xtile, stage_voff(piece * 64 + lane), smem_a + (unsigned)(Cfg::kStageX + piece * 1024));
                }
            }
            ACX_FENCE
            if (u + 1 < Cfg::kUnits) ACX_TOUCH1(f[(u + 1) % 3])
        }
        if constexpr (HV) Xv1 = Xn[1];
        ACX_NS_FINE()
        ACX_SEG_END(XD == 2)
    };

    using T_ = std::true_type; using F_ = std::false_type;
    using XD0 = std::integral_constant<int, 0>; using XD1 = std::integral_constant<int, 1>; using XD2 = std::integral_constant<int, 2>;
    ACX_SB_LOAD(0)
#pragma unroll
    for (int j = 0; j < 16; ++j) { ACX_BIAS_SEL(j) }
    phase1(F_{}, 0, 0);
    ACX_NS_STAMP()      // 4
    // the first chunk has no phase 2 to ride on: tile 0's GELU and the bias of chunk 1 stand alone
    ACX_MICRO_RANGE(Xn[0], 0, 0, Cfg::kGeluSteps)
    Xv1 = Xn[1];
    ACX_SB_LOAD(1)
#pragma unroll
    for (int j = 0; j < 16; ++j) { ACX_BIAS_SEL(j) }
    ACX_NS_STAMP()      // 5
    int slot = 1;
    for (int k = 1; k < n - 1; ++k) {
        phase1(T_{}, 2 * k - 1, slot);                       // X(k); GELU of tile 1 of chunk k - 1, G(k - 1) written
        ACX_NS_STAMP()  // 6 + 2 (k - 1)
        slot = slot == 2 ? 0 : slot + 1;
        phase2(T_{}, T_{}, XD0{}, k + 1, 2 * k, slot);       // out += W2(k - 1) G(k - 1); GELU of tile 0 of chunk k; bias of chunk k + 1
        ACX_NS_STAMP()  // 7 + 2 (k - 1)
        slot = slot == 2 ? 0 : slot + 1;
    }
    ACX_NS_STAMP()      // 6 + 2 (n - 2): the loop has ended
    phase1(T_{}, 2 * n - 3, slot);                           // X(n - 1); G(n - 2) written
    slot = slot == 2 ? 0 : slot + 1;
    phase2(T_{}, F_{}, XD1{}, 0, 2 * n - 2, slot);           // slot 1; x rows 0..63 -> slot 0
    slot = slot == 2 ? 0 : slot + 1;
    ACX_MICRO_RANGE(Xv1, 1, 0, Cfg::kGeluSteps)             // tile 1 of the last chunk: no phase 1 left to ride on
    ACX_WRITE_G()             // nobody reads the buffer between the barrier above and the one below
    ACX_LDS_BARRIER
    phase2(F_{}, F_{}, XD2{}, 0, 2 * n - 1, slot);           // slot 2; x rows 64..127 -> slot 1
    ACX_NS_STAMP()      // 7 + 2 (n - 2): the last three segments + the G hand-over

    // ---- epilogue: out += I . x (x read from the staged rows as B fragments), round, rows back through the LDS ----
    const int cr = (C / 16) * r;          // first 16-byte chunk of this wave's channel half in a row
    {
        f32x4 ident[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const int e = l31 - 16 * sp - 8 * hh;             // this lane's row of I has its 1 at k = 16 s' + 8 hh + e
            unsigned w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) w[p] = (e == 2 * p ? 0x3f80u : 0u) | (e == 2 * p + 1 ? 0x3f800000u : 0u);
            ident[sp] = __builtin_bit_cast(f32x4, uint4{w[0], w[1], w[2], w[3]});
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rowl = i == 0 ? rown : rowp;
            const char* rp = smem + Cfg::kStageX + rowl * RB;
            const int sx = Cfg::swz1(rowl);
#pragma unroll
            for (int t = 0; t < Cfg::kTW; ++t)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const f32x4 xf = *reinterpret_cast<const f32x4*>(rp + (((cr + 4 * t + 2 * sp + hh) ^ sx) << 4));
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ACX_B8(ident[sp]), ACX_B8(xf), acc[i][t], 0, 0, 0);
                }
        }
    }
    ACX_NS_STAMP()      // + 1: residual added
    if constexpr (!LNOUT) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rowl = i == 0 ? rown : rowp;
            char* rp = smem + Cfg::kStageX + rowl * RB;
            const int sx = Cfg::swz1(rowl);
#pragma unroll
            for (int t = 0; t < Cfg::kTW; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // lanes (px, 0) and (px, 1) trade pieces so that each holds 8 consecutive channels: 32 t + 16 j + 8 hh .. + 7
                    unsigned e[2], o[2];
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        e[w] = ns_pack_bf16(acc[i][t][8 * j + 2 * w], acc[i][t][8 * j + 2 * w + 1]);
                        o[w] = ns_pack_bf16(acc[i][t][8 * j + 4 + 2 * w], acc[i][t][8 * j + 4 + 2 * w + 1]);
                        acx_pair_swap(e[w], o[w]);
                    }
                    *reinterpret_cast<uint4*>(rp + (((cr + 4 * t + 2 * j + hh) ^ sx) << 4)) = uint4{e[0], e[1], o[0], o[1]};
                }
        }
        // this wave's quarter of the tile -- the pair's 64 rows, its channel half -- leaves as whole half rows: 16 bytes per lane,
        // C/16 lanes per row (same-wave LDS operations execute in order: no wait between the writes above and these reads)
        char* xo = reinterpret_cast<char*>(x) + row0 * RB;
        auto store_rows = [&](auto masked) __attribute__((always_inline)) {
            constexpr bool kMasked = decltype(masked)::value;
#pragma unroll
            for (int q = 0; q < C / 16; ++q) {
                const int idx = q * 64 + lane;
                const int rr = idx / (C / 16), cc = cr + (idx - rr * (C / 16));
                const int rowl = 64 * ph + rr;
                const uint4 v = *reinterpret_cast<const uint4*>(smem + Cfg::kStageX + rowl * RB + ((cc ^ Cfg::swz1(rowl)) << 4));
                if (!kMasked || rowl < rows_valid) *reinterpret_cast<uint4*>(xo + rowl * RB + cc * 16) = v;
            }
        };
        if (rows_valid == Cfg::kPix) store_rows(std::false_type{});
        else store_rows(std::true_type{});
    }
    ACX_NS_STAMP()      // + 2: stores issued
#undef ACX_WDMA
#undef ACX_WDMA_LOOP
#undef ACX_MICRO_AFTER
#undef ACX_MICRO_HEAD
#undef ACX_WRITE_G
#undef ACX_SB_LOAD
#undef ACX_SB_TOUCH
#undef ACX_BIAS_SEL
#undef ACX_SEG_WAIT_N
#undef ACX_SEG_BAR
#undef ACX_LDS_BARRIER
#undef ACX_B8
#undef ACX_FENCE
#undef ACX_W1_RD
#undef ACX_W2_RD
#undef ACX_G_RD
#undef ACX_MICRO_RANGE
#undef ACX_MFMA
#undef ACX_TOUCH1
#undef ACX_SEG_END
}

bool mlp_nsplit_bf16_supported(int C, bool ln_out) { return C == 384 && !ln_out; }

int launch_mlp_nsplit_bf16(acx_ctx* c, const BlockW& w, int C, const void* y, void* x, long long M, hipStream_t s) {
    if (!w.wstream_b || !w.b1) ACX_FAIL(ACX_ERR_STATE, "N-split fused bf16 MLP: the weight stream was not packed for C=%d", C);
    if (C != 384) ACX_FAIL(ACX_ERR_SHAPE, "N-split fused bf16 MLP: unsupported channel count %d", C);
    using Cfg = NsCfg<384>;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "ring + G exchange do not fit the LDS");
    ProfScope ps(c, ACX_K_MLP_WIDE, s);
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_nsplit_bf16_kernel<384, false>, kCuLdsBytes));
    const long long blocks = (M + Cfg::kPix - 1) / Cfg::kPix;
    launch_kernel(&mlp_nsplit_bf16_kernel<384, false>, dim3((unsigned)blocks), dim3(Cfg::kThreads), kCuLdsBytes /* CU-exclusive */, s,
        reinterpret_cast<const __bf16*>(y), reinterpret_cast<__bf16*>(x), reinterpret_cast<const char*>(w.wstream_b), w.b1, w.b2, M, 0,
        static_cast<__bf16*>(nullptr));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

}  // namespace acx
