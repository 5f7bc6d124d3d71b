// K4fw -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86) in
// split-fp16 arithmetic for WIDE stages (C = 384): the hidden activation (B H W x 4C, 347 MB per block at B = 64) never
// exists in memory (hard part 1).  Same transposed dataflow as mlp_fused_split.hip -- a wave owns 32 pixels,
// the accumulator tile of the first product is the B operand of the second -- re-planned for one wave per SIMD.
//
// Matrix instruction: v_mfma_f32_16x16x32_f16 (since round 3; 32x32x16 before).  Same flops per cycle, but under MFMA load
// the chip holds a higher clock on this shape (MI355X_MICROARCH.md 'DVFS give-back' item 7; profiles/r03_i_mfma_shape_rates.txt;
// same-box A/B of this kernel: -3 % per launch, -2.7 % per forward).  A wave's 32 pixels are two blocks of 16 (pb), a chunk's
// 32 hidden units two blocks of 16 (hb); lane (l15, g4) is row / column l15 and k block g4 (8 of the 32 k) of every operand:
//   phase 1   X^T block (hb, pb) [16 hidden x 16 px] += W1c rows 16 hb .. (one fragment, 32 channels) . LN(y)^T block pb
//   phase 2   out^T block (cb, pb) [16 channels x 16 px] += W2c rows 16 cb .. (one fragment, the chunk's 32 hidden units) . G(pb)
// A weight fragment feeds both pixel blocks; an accumulator block is 4 registers: lane holds rows 4 g4 .. + 3 of column l15,
// so the 8 pre-activations of a lane's two X blocks of one pixel block are, as they stand, its 8 k values of phase 2's B
// operand (k slot 8 g4 + j = hidden unit 16 (j >> 2) + 4 g4 + (j & 3): acx_finalize stores W2c's columns in that order).
//
//   registers   the workgroup is CU-exclusive (acx_internal.h): 4 waves x 512 registers.  Per lane: C/2 for the wave's
//               normalised activations (hi + lo halves, B operand of phase 1), C/2 accumulators of out^T, ~100 working.
//   weights     acx_finalize lays every block's weights out as ONE stream of segments in consumption order,
//                 W1(0) W1(1) W2(0) W1(2) W2(1) ... W1(n-1) W2(n-2) W2(n-1)        (n = 4C/32 hidden chunks)
//               a segment = the [32 x C] (W1) or [C x 32] (W2) S16 image of one chunk, 128 C bytes, already in LDS image
//               order (XOR swizzle baked in), so the LDS-DMA source of a piece is base + 1 KB x piece + 16 x lane.
//               LDS holds a ring of three segments: one being multiplied, one landed or landing, one being requested.
//   schedule    segment 2k-1: phase 1 of chunk k    X^T[32 hidden x 32 px] = W1c . LN(y)^T     3 C/16 MFMA pairs (one per pixel block)
//               segment 2k  : phase 2 of chunk k-1  out^T[C x 32 px] += W2c . G(k-1)            3 C/16 MFMA pairs
//               GELU + hi/lo split of X(k) -> G(k): 21 single-instruction "nano-steps" per register pair (split_math.h,
//               gelu3_nano), half of the pairs dealt over the MFMAs of segment 2k, the other half over those of segment
//               2k+1.  One wave per SIMD issues both streams: about five single-issue instructions ride for free behind
//               32 cycles of matrix work (a 32x32x16 MFMA then, a pair of 16x16x32 now), every further one costs ~5 cycles
//               (profiles/r03_d_coissue_table_full.txt), so each gap behind a pair gets at most four slots, fewer where an LDS-DMA piece or the fragment reads of the next unit already
//               sit in it (WideCfg::gap_free).
//               The DMA pieces of segment s+2 are threaded through the units of segment s; one counted s_waitcnt vmcnt +
//               s_barrier per segment.
// HBM traffic per block: read y, read x, write x (3 C H W 4 bytes) + the weight stream from L2 / MALL.
#include <type_traits>

#include <cstdlib>
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
    static constexpr size_t kLdsBytes = 3 * (size_t)kSegBytes + 4 * C * 4 + C * 4;      // ring, b1 (pre-divided), b2
    static constexpr int kMfmas = 3 * kUnits * PT;          // MFMAs per segment
    static constexpr int kDmaStride = kUnits / kPieces;     // one LDS-DMA piece every kDmaStride units
    // GELU nano-steps (single instructions, split_math.h) a segment carries: half of a pixel tile's 8 register pairs
    static constexpr int kNano = 4 * kGelu3Nano * PT;
    static_assert(C % 32 == 0 && kSteps == kUnits && kUnits % kPieces == 0, "unit / piece bookkeeping");
    // Issue budget of the gap BEHIND MFMA m of a segment (unit m / 6, position m % 6: the three terms x the two pixel blocks).
    // Behind a 16x16x32 MFMA (16 matrix cycles) TWO single-issue instructions are free and every further one costs ~5 cycles;
    // fillers behind the second MFMA of a back-to-back pair do not use the first one's shadow (tools/lab/coissue2.hip,
    // profiles/r03_s_coissue_16x16.txt: pair + 4 fillers 44.0 cycles per 32 matrix cycles, 2 behind each MFMA 34.3).  So every
    // filler of the loop has a gap of its own: position 0 the LDS-DMA piece of every kDmaStride-th unit (one instruction),
    // position 1 the read of the lo fragment three units ahead (its register is dead after the first term), position 4 the
    // counted wait for the next unit's fragments, position 5 the read of the hi fragment; what is left -- 2 / 1 / 2 / 2 / 1 / 1 --
    // is dealt to the GELU in proportion.
    // (closed forms, no loops: the arguments become constants only after the unit loops are unrolled, and everything derived
    // from them -- register indices above all -- must fold then)
    __host__ __device__ static constexpr int cum_cap_units(int u) { return 9 * u - ((u + kDmaStride - 1) / kDmaStride); }   // gaps of units [0, u)
    __host__ __device__ static constexpr int cum_cap(int m) {       // free slots of gaps [0, m]
        const int u = m / 6, pos = m % 6;
        const int c0 = (u % kDmaStride == 0) ? 1 : 2;
        return cum_cap_units(u) + (pos == 0 ? c0 : pos == 1 ? c0 + 1 : pos == 2 ? c0 + 3 : pos == 3 ? c0 + 5 : pos == 4 ? c0 + 6 : c0 + 7);
    }
    // what does not fit the gaps is issued at the top of the segment, right behind the reads of its first three fragment pairs
    // (nothing since the GELU is 21 steps per pair: 84 per segment against 102 slots at C = 192; it was 120 with the 30-step form)
    static constexpr int kCapTotal = cum_cap_units(kUnits);
    static constexpr int kNanoHead = kNano > kCapTotal ? ((kNano - kCapTotal + 1) / 2) * 2 : 0;
    __host__ __device__ static constexpr int nano_end(int m) {      // nano-steps issued once the gap behind MFMA m is done
        return kNanoHead + ((kNano - kNanoHead) * cum_cap(m) + kCapTotal / 2) / kCapTotal;
    }
    __host__ __device__ static constexpr int nano_begin(int m) { return m == 0 ? kNanoHead : nano_end(m - 1); }
    // W1 rows are 4 C bytes: the XOR that spreads 16 consecutive rows over the LDS banks (api.hip packs the image with it)
    __device__ static int swz1(int row) { return (C % 64 == 0) ? (row & 15) : ((row >> 1) & 7); }
};

// byte offset of segment s in the stream: order W1(0) W1(1) W2(0) W1(2) W2(1) ... (see the header)
//   W1(k): k == 0 ? 0 : 2k - 1       W2(k): k == n - 1 ? 2n - 1 : 2k + 2
// NPB = 16-pixel blocks per wave: 2 (a 128-pixel tile per workgroup), or 1 for launches whose 64-pixel tiles all find a CU at
// once (small batches: twice the workgroups, ~0.65 of the time each; the arithmetic of a pixel is the same in both)
// PERS (round 4, C = 192): ONE workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...  A CU-exclusive workgroup cannot hide
// its own HBM round trips behind a neighbour's arithmetic, and with every CU starting its tile at the same moment the y-row
// loads of the prologue and the stores of the epilogue arrive at the memory system as two bursts per round (tools/lab:
// prologue 14 k + epilogue 11 k of a tile's 111 k cycles at C = 192, profiles/r04_k_wide192_ablation.txt).  The persistent
// form requests the NEXT tile's y rows into registers the C = 192 kernel has to spare (96 per lane) during segment 2n - 5, lets
// the weight ring run on across the tile boundary (segments 0 and 1 of the next tile are requested during the last two), and
// leaves the stores of a tile in flight under the LayerNorm and the first segment of the next one.  C = 384 has no registers
// left for the rows and keeps the one-tile form.
template <int C, int PT, bool LNOUT, int NPB, bool PERS>
__global__ __launch_bounds__(256) void mlp_fused_wide_kernel(
    const float* __restrict__ y, float* __restrict__ x, const char* __restrict__ wstream /*[2n][128 C bytes]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, float sinv1, float sinv2, float hscale,
    char* __restrict__ ln_out /* LNOUT: (M, C) S16 rows of LayerNorm(x_new) x 2^11, written INSTEAD of x */) {
    using Cfg = WideCfg<C, PT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + 3 * Cfg::kSegBytes);   // [4C], pre-divided by sinv1
    float* b2s = b1s + 4 * C;                                           // [C]: read by the epilogue (as loads from b2 in its
    // loop they were one L2 round trip per 16 channels -- hipcc waits vmcnt(0) for each, i.e. for the previous store as well)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert(PT == 1, "the 16x16x32 form is written for one 32-pixel tile per wave");
    const int l15 = lane & 15, g4 = lane >> 4;       // 16x16x32 MFMA: lane = (row / column l15, k block g4)
    ACX_CLAIM_VGPR(255);          // CU-exclusive: one wave per SIMD holds the SIMD's whole register file
    ACX_CLAIM_AGPR(255);
    ACX_WSTAMP_DECL
    ACX_WSTAMP(-1)
    long long mrow[2];            // this lane's pixel row in each of the wave's two 16-pixel blocks
    bool valid[2];
    static_assert(!PERS || NPB == 2, "the persistent form is written for 128-pixel tiles");
    constexpr int kPixT = Cfg::kWaves * 16 * NPB;
    const long long n_tiles = (M + kPixT - 1) / kPixT;
    long long tile = blockIdx.x;
#define ACX_TILE_ROWS(t_)                                                                                       \
    _Pragma("unroll") for (int pb = 0; pb < 2; ++pb) {                                                          \
        mrow[pb] = (t_) * kPixT + wave * (16 * NPB) + (pb < NPB ? pb : 0) * 16 + l15;                           \
        valid[pb] = pb < NPB && mrow[pb] < M;                                                                   \
        if (mrow[pb] >= M) mrow[pb] = M - 1;                                                                    \
    }
    ACX_TILE_ROWS(tile)

    constexpr int n = Cfg::kChunks;
    // (issued from inline asm: next to the builtin form hipcc waits lgkmcnt(0) / vmcnt(0) wherever an LDS read follows, which
    // defeats the two-unit look-ahead of the fragment reads; the counted waits at the segment ends are this file's own)
    const unsigned smem_a = acx_lds_addr(smem);
    // pieces 8 g .. 8 g + 7 of a segment share one M0 / base pair, set at the group's first piece and centred on its fifth
    // (round 5: the base is a SCALAR -- the wave's first piece of the run -- and the lane contributes 16 x lane as a 32-bit offset:
    // split_math.h acx_glds16_run_s; the 64-bit address per lane of the earlier form made a piece's issue several times dearer)
    const char* wbase = wstream;
    const unsigned dma_voff = lane * 16;
#define ACX_WDMA(seg_, piece_, grp_)                                                                             \
        {   if ((piece_) % 8 == 0) {                                                                             \
                wbase = wstream + (long long)(seg_) * Cfg::kSegBytes + (wave * Cfg::kPieces + (piece_) + 4) * 1024;      \
                acx_set_m0(__builtin_amdgcn_readfirstlane(smem_a + (grp_) * Cfg::kSegBytes + (wave * Cfg::kPieces + (piece_) + 4) * 1024)); \
            }                                                                                                    \
            acx_glds16_run_s(wbase, dma_voff, (piece_) % 8); }
    // segments 0 and 1 are requested before anything else; segment s + 2 follows during segment s
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(0, p, 0)
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) ACX_WDMA(1, p, 1)
    // bias staging: 16 bytes per lane, requested here (ahead of the tile's rows: vmcnt completes in order) and written to the LDS
    // after the rows have been normalised -- no wait of its own (a rolled scalar loop in this place was 4C / 256 dependent round
    // trips in front of the row loads)
    constexpr int kB1Iters = (C + Cfg::kThreads - 1) / Cfg::kThreads;
    float4 b1v[kB1Iters];
#pragma unroll
    for (int q = 0; q < kB1Iters; ++q) {
        const int i = q * Cfg::kThreads + tid;
        b1v[q] = reinterpret_cast<const float4*>(b1)[i < C ? i : 0];
    }
    const float4 b2v = reinterpret_cast<const float4*>(b2)[tid < C / 4 ? tid : 0];

    // ---- this wave's activations: lane (px = l15 of block pb, k block g4) holds channels 32 s + 8 g4 .. + 7, s = 0..C/32-1 ----
    constexpr int kS32 = C / 32;
    f32x4 acth[2][kS32], actl[2][kS32];                         // 8 fp16 halves each
    // PERS: the rows of a tile are requested one tile ahead, through a pointer the compiler knows nothing about (as loads of a
    // const __restrict__ argument they could be moved anywhere, e.g. down to their first use); the counted waits of the
    // segments around the request allow for exactly kAhead further vector-memory instructions in flight (ACX_SEG_END)
    constexpr int kAhead = 2 * (C / 16);                        // loads of the next tile's y rows = loads of x = stores of x
    typedef const float __attribute__((address_space(1))) * GlobalF;       // (laundered pointers lose the inferred address space)
    typedef const f32x4 __attribute__((address_space(1))) * GlobalF4;
    GlobalF yl = (GlobalF)y;
    if constexpr (PERS) asm volatile("" : "+s"(yl));
    f32x4 yraw[PERS ? 2 : 1][PERS ? 2 * kS32 : 1];
    long long mrow_n[2];
#define ACX_Y_LOAD(rows_)                                                                                       \
    _Pragma("unroll") for (int pb = 0; pb < 2; ++pb) {                                                          \
        GlobalF yp = yl + (rows_)[pb] * C + 8 * g4;                                                             \
        _Pragma("unroll") for (int s = 0; s < kS32; ++s) {                                                      \
            yraw[pb][2 * s] = *(GlobalF4)(yp + 32 * s);                                                         \
            yraw[pb][2 * s + 1] = *(GlobalF4)(yp + 32 * s + 4);                                                 \
        }                                                                                                       \
    }
    // bias staging: 16 bytes per lane into the LDS (the loads were requested ahead of the tile's rows: vmcnt completes in order)
#define ACX_BIAS_STAGE()                                                                                        \
    {                                                                                                           \
        const float b1scale = 1.0f / sinv1;             /* a power of two */                                    \
        _Pragma("unroll") for (int q = 0; q < kB1Iters; ++q) {                                                  \
            const int i = q * Cfg::kThreads + tid;                                                              \
            float4 v = b1v[q];                                                                                  \
            v.x *= b1scale; v.y *= b1scale; v.z *= b1scale; v.w *= b1scale;                                     \
            if (i < C) reinterpret_cast<float4*>(b1s)[i] = v;                                                   \
        }                                                                                                       \
        static_assert(C / 4 <= Cfg::kThreads, "one float4 of b2 per thread");                                   \
        if (tid < C / 4) reinterpret_cast<float4*>(b2s)[tid] = b2v;                                             \
    }
    if constexpr (PERS) { ACX_Y_LOAD(mrow) ACX_BIAS_STAGE() }
    bool first = true;
    for (;;) {                                                  // PERS: tiles blockIdx.x, + gridDim.x, ...; else one pass
    const bool has_next = PERS && tile + gridDim.x < n_tiles;
    if constexpr (PERS) {
        const long long tn = has_next ? tile + gridDim.x : tile;     // (no successor: the rows are requested all the same -- the waits count them)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            mrow_n[pb] = tn * kPixT + wave * (16 * NPB) + pb * 16 + l15;
            if (mrow_n[pb] >= M) mrow_n[pb] = M - 1;
        }
    }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        float a[C / 4];
        const float* yp = y + mrow[pb] * C + 8 * g4;
#pragma unroll
        for (int s = 0; s < kS32; ++s) {
            f32x4 v0, v1;
            if constexpr (PERS) { v0 = yraw[pb][2 * s]; v1 = yraw[pb][2 * s + 1]; }
            else { v0 = *reinterpret_cast<const f32x4*>(yp + 32 * s); v1 = *reinterpret_cast<const f32x4*>(yp + 32 * s + 4); }
            a[8 * s + 0] = v0.x; a[8 * s + 1] = v0.y; a[8 * s + 2] = v0.z; a[8 * s + 3] = v0.w;
            a[8 * s + 4] = v1.x; a[8 * s + 5] = v1.y; a[8 * s + 6] = v1.z; a[8 * s + 7] = v1.w;
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < C / 4; ++i) sum += a[i];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < C / 4; ++i) { const float t = a[i] - mean; d = fmaf(t, t, d); }
        d += __shfl_xor(d, 16);
        d += __shfl_xor(d, 32);
        const float sc = kSplitLnScale / sqrtf(d * (1.0f / C) + 1e-6f);
#pragma unroll
        for (int s = 0; s < kS32; ++s) {
            unsigned uh4[4], ul4[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                acx_split_pair((a[8 * s + 2 * p] - mean) * sc, (a[8 * s + 2 * p + 1] - mean) * sc, uh4[p], ul4[p]);
            }
            acth[pb][s] = __builtin_bit_cast(f32x4, uint4{uh4[0], uh4[1], uh4[2], uh4[3]});
            actl[pb][s] = __builtin_bit_cast(f32x4, uint4{ul4[0], ul4[1], ul4[2], ul4[3]});
        }
    }

    if constexpr (!PERS) { ACX_BIAS_STAGE() }     // behind the rows, so that it has no wait of its own
    f32x4 acc[C / 16][2];         // out^T: block cb = 16 out channels x pixel block pb; lane holds channels 16 cb + 4 g4 .. + 3
#pragma unroll
    for (int cb = 0; cb < C / 16; ++cb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a segment (without the ring offset) for v_mfma_f32_16x16x32_f16: lane (l15, g4) reads row
    // l15 of a 16-row block, the 8 k of block g4 of the 32-k step:
    //   W1 image: row = hidden unit (4 C bytes = C/4 chunks of 16 B), 16-row block hb; chunk p = 8 s32 + 2 g4 + pl at position
    //             p ^ l15 (pl = 0 hi halves, 1 lo halves; the XOR touches the low 4 bits only: C % 64 == 0)
    //   W2 image: row = out channel (128 B = 8 chunks), block cb rows 16 cb + l15, chunk 2 g4 + pl at position ^ acx_swz8(l15)
    static_assert(C % 64 == 0, "W1 image swizzle");
    int w1off[2][2], w2off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w1off[q][pl] = l15 * (4 * C) + (((8 * q + 2 * g4 + pl) ^ l15) << 4);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) w2off[pl] = l15 * 128 + (((2 * g4 + pl) ^ acx_swz8(l15)) << 4);
    const GeluK3 gk = gelu_k3(sinv1, hscale);

#define ACX_H8(v_) __builtin_bit_cast(h8, v_)
#define ACX_FENCE __builtin_amdgcn_sched_barrier(0);
    // phase-1 unit u_ = (32-channel step s32 = u_ >> 1, hidden block hb = u_ & 1): the bits of the chunk index above the XORed
    // four = s32 >> 1 -> + 256 B each
#define ACX_W1_RD(base_, u_, pl_) (*reinterpret_cast<const f32x4*>((base_) + ((u_) & 1) * (16 * 4 * C) + ((u_) >> 2) * 256 + w1off[((u_) >> 1) & 1][pl_]))
    // MFMA m_ of a segment is followed (behind a scheduling fence) by its share of the GELU nano-steps the segment carries
    // (WideCfg::nano_begin / nano_end: at most two, fewer where a tenant sits in the gap)
#define ACX_NANO_AT(HV_, half_, m_)                                                                             \
        ACX_FENCE if constexpr (HV_) { ACX_NANO_RANGE(half_, Cfg::nano_begin(m_), Cfg::nano_end(m_)) } ACX_FENCE
#define ACX_M16(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ACX_H8(a_), ACX_H8(b_), c_, 0, 0, 0);
    // phase-2 unit i_ = block cb of 16 out channels; K = the chunk's 32 hidden units in ONE step
#define ACX_W2_RD(base_, i_, pl_) (*reinterpret_cast<const f32x4*>((base_) + (i_) * (16 * 128) + w2off[pl_]))
    // nano-steps [from, to) of the kNano that segment half half_ carries: step ng_ is instruction (ng_ % 42) / 2 of register pair
    // 4 half_ + 2 (ng_ / 42) + (ng_ & 1) (split_math.h, gelu3_nano): pair after pair, in order.  Pair pr = 4 pb + 2 hb + e covers registers
    // 2 e, 2 e + 1 of X block 2 pb + hb: hidden units 16 hb + 4 g4 + 2 e, + 1 of pixel block pb
#define ACX_NANO_ARGS gsv[w_], gk, Xv[(2 * pr_) >> 2][(2 * pr_) & 3], Xv[(2 * pr_) >> 2][((2 * pr_) & 3) + 1], uh[0][pr_], ul[0][pr_]
#define ACX_NANO_CASE(I_) else if (st_ == (I_)) gelu3_nano<(I_)>(ACX_NANO_ARGS);
#define ACX_NANO_RANGE(half_, from_, to_)                                                                       \
        _Pragma("unroll") for (int ng_ = (from_); ng_ < (to_); ++ng_) {                                         \
            /* two register pairs in flight, their instructions alternating: consecutive nano-steps never depend on each other */ \
            const int j_ = ng_ % (2 * kGelu3Nano), w_ = j_ & 1;                                                 \
            const int pr_ = 4 * (half_) + 2 * (ng_ / (2 * kGelu3Nano)) + w_, st_ = j_ >> 1;                     \
            if (st_ == 0) gelu3_nano<0>(ACX_NANO_ARGS);                                                         \
            ACX_NANO_CASE(1)                                                                                     \
            ACX_NANO_CASE(2)                                                                                     \
            ACX_NANO_CASE(3)                                                                                     \
            ACX_NANO_CASE(4)                                                                                     \
            ACX_NANO_CASE(5)                                                                                     \
            ACX_NANO_CASE(6)                                                                                     \
            ACX_NANO_CASE(7)                                                                                     \
            ACX_NANO_CASE(8)                                                                                     \
            ACX_NANO_CASE(9)                                                                                     \
            ACX_NANO_CASE(10)                                                                                    \
            ACX_NANO_CASE(11)                                                                                    \
            ACX_NANO_CASE(12)                                                                                    \
            ACX_NANO_CASE(13)                                                                                    \
            ACX_NANO_CASE(14)                                                                                    \
            ACX_NANO_CASE(15)                                                                                    \
            ACX_NANO_CASE(16)                                                                                    \
            ACX_NANO_CASE(17)                                                                                    \
            ACX_NANO_CASE(18)                                                                                    \
            ACX_NANO_CASE(19)                                                                                    \
            ACX_NANO_CASE(20)                                                                                    \
        }
#define ACX_TOUCH2(h_, l_) asm volatile("" :: "v"(h_), "v"(l_));
#define ACX_BIAS_INIT(j_)                                                                                       \
        _Pragma("unroll") for (int hb_ = 0; hb_ < 2; ++hb_) {                                                   \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * (j_) + 16 * hb_ + 4 * g4);              \
            Xn[hb_] = bq; Xn[2 + hb_] = bq;                                                                     \
        }
    // G of the chunk as the B operand of phase 2: pixel block pb takes pairs 4 pb .. + 3 -- k slot 8 g4 + j of the MFMA is
    // hidden unit 16 (j >> 2) + 4 g4 + (j & 3) of the chunk; acx_finalize stores W2c's columns in that order
#define ACX_PACK_G()                                                                                            \
        _Pragma("unroll") for (int pb_ = 0; pb_ < 2; ++pb_) {                                                   \
            gh[pb_] = __builtin_bit_cast(f32x4, uint4{uh[0][4 * pb_ + 0], uh[0][4 * pb_ + 1], uh[0][4 * pb_ + 2], uh[0][4 * pb_ + 3]}); \
            gl[pb_] = __builtin_bit_cast(f32x4, uint4{ul[0][4 * pb_ + 0], ul[0][4 * pb_ + 1], ul[0][4 * pb_ + 2], ul[0][4 * pb_ + 3]}); }
    // end of a segment: the pieces requested during it may stay in flight, everything older must have landed, and
    // every wave must be done reading the segment before its ring slot is requested again
    // (extra_: PERS -- kAhead loads or stores issued since the pieces of the PREVIOUS segment may stay in flight as well; they
    // are younger than those pieces, and vector-memory operations complete in order)
#define ACX_SEG_END(issued_, stamp_, extra_)                                                                    \
        ACX_FENCE                                                                                               \
        ACX_WSTAMP(stamp_)                                                                                      \
        if (issued_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::kPieces + (extra_)) : "memory");            \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                           \
        ACX_FENCE                                                                                               \
        ACX_WSTAMP(3)

    f32x4 Xn[4], Xv[4];           // X^T blocks 2 pb + hb (16 hidden x 16 pixels each): Xn being accumulated by phase 1; Xv: the previous chunk's, input of the GELU
    f32x4 gh[2], gl[2];           // G(k - 1): B operand of phase 2 per pixel block, hi / lo halves
    unsigned uh[PT][8], ul[PT][8];
    GeluState3 gsv[2];

    // one phase-1 segment: X = b1 + W1c . LN(y)^T for chunk k_, image in ring slot grp_, requesting segment seg_ + 2
    // opt (PERS): bit 0 = the segment-end wait leaves kAhead further operations in flight; bit 1 = request the next tile's rows
    auto phase1 = [&](auto with_gelu, const int k_, const int seg_, const int grp_, auto opt) __attribute__((always_inline)) {
        constexpr int kOpt = decltype(opt)::value;
        // second half of the GELU of Xv (the pairs of pixel block 1) rides on this segment's MFMAs
        constexpr bool HV = decltype(with_gelu)::value && NPB == 2;
        constexpr bool kPack = decltype(with_gelu)::value;
        const char* base = smem + grp_ * Cfg::kSegBytes;
        const bool dma = seg_ + 2 < Cfg::kSegs || has_next;
        const int dseg = seg_ + 2 < Cfg::kSegs ? seg_ + 2 : seg_ + 2 - Cfg::kSegs;      // the ring runs on into the next tile
        const int g2 = (grp_ + 2) % 3;
        ACX_BIAS_INIT(k_)
        // fragments are read TWO units ahead of their MFMAs (three register sets rotating): with one unit of look-ahead
        // (96 matrix cycles) the LDS latency under load was exposed in front of every unit
        static_assert(Cfg::kSteps % 3 == 0, "the unit loop is unrolled by three");
        f32x4 f0h = ACX_W1_RD(base, 0, 0), f0l = ACX_W1_RD(base, 0, 1), f1h = ACX_W1_RD(base, 1, 0), f1l = ACX_W1_RD(base, 1, 1),
              f2h = ACX_W1_RD(base, 2, 0), f2l = ACX_W1_RD(base, 2, 1);
        ACX_FENCE
        if constexpr (HV) { ACX_NANO_RANGE(1, 0, Cfg::kNanoHead) }
        ACX_FENCE
        // unit u_ = (s32 = u_ >> 1, hb = u_ & 1), fragments (ch_, cl_); X block index = 2 pb + hb.  Terms: lo x hi, hi x lo, hi x hi,
        // each over the two pixel blocks.  Every filler has its own gap (WideCfg): the lo register is dead after the first
        // term -- the lo fragment of unit u_ + 3 is read into it there; the hi register after the last MFMA.
#define ACX_P1_UNIT(u_, ch_, cl_, th_, tl_)                                                                     \
            ACX_FENCE                                                                                           \
            ACX_M16(cl_, acth[0][(u_) >> 1], Xn[(u_) & 1])                                                      \
            if ((u_) % Cfg::kDmaStride == 0 && dma) { ACX_WDMA(dseg, (u_) / Cfg::kDmaStride, g2) }              \
            ACX_NANO_AT(HV, 1, 6 * (u_) + 0)                                                                    \
            if constexpr (NPB == 2) { ACX_M16(cl_, acth[1][(u_) >> 1], Xn[2 + ((u_) & 1)]) }                        \
            if ((u_) + 3 < Cfg::kSteps) cl_ = ACX_W1_RD(base, (u_) + 3, 1);                                     \
            ACX_NANO_AT(HV, 1, 6 * (u_) + 1)                                                                    \
            ACX_M16(ch_, actl[0][(u_) >> 1], Xn[(u_) & 1])                                                      \
            ACX_NANO_AT(HV, 1, 6 * (u_) + 2)                                                                    \
            if constexpr (NPB == 2) { ACX_M16(ch_, actl[1][(u_) >> 1], Xn[2 + ((u_) & 1)]) }                        \
            ACX_NANO_AT(HV, 1, 6 * (u_) + 3)                                                                    \
            ACX_M16(ch_, acth[0][(u_) >> 1], Xn[(u_) & 1])                                                      \
            if ((u_) + 1 < Cfg::kSteps) ACX_TOUCH2(th_, tl_)                                                    \
            ACX_NANO_AT(HV, 1, 6 * (u_) + 4)                                                                    \
            if constexpr (NPB == 2) { ACX_M16(ch_, acth[1][(u_) >> 1], Xn[2 + ((u_) & 1)]) }                        \
            if ((u_) + 3 < Cfg::kSteps) ch_ = ACX_W1_RD(base, (u_) + 3, 0);                                     \
            ACX_NANO_AT(HV, 1, 6 * (u_) + 5)                                                                    \
            ACX_FENCE
#pragma unroll
        for (int s = 0; s < Cfg::kSteps; s += 3) {
            ACX_P1_UNIT(s, f0h, f0l, f1h, f1l)
            ACX_P1_UNIT(s + 1, f1h, f1l, f2h, f2l)
            ACX_P1_UNIT(s + 2, f2h, f2l, f0h, f0l)
        }
#undef ACX_P1_UNIT
        if constexpr (kPack) { ACX_PACK_G() }
#pragma unroll
        for (int q = 0; q < 4; ++q) Xv[q] = Xn[q];
        if constexpr ((kOpt & 2) != 0) { ACX_FENCE ACX_Y_LOAD(mrow_n) }
        ACX_SEG_END(dma, 1, (kOpt & 1) ? kAhead : 0)
    };
    // one phase-2 segment: out^T += W2c . G for the chunk whose G sits in gh / gl, image in ring slot grp_; with_gelu:
    // the first half of the GELU + split of Xv (the NEXT chunk) rides on this segment's MFMAs
    auto phase2 = [&](auto with_gelu, const int seg_, const int grp_, auto opt) __attribute__((always_inline)) {
        constexpr bool HV = decltype(with_gelu)::value;
        constexpr int kOpt = decltype(opt)::value;
        const char* base = smem + grp_ * Cfg::kSegBytes;
        const bool dma = seg_ + 2 < Cfg::kSegs || has_next;
        const int dseg = seg_ + 2 < Cfg::kSegs ? seg_ + 2 : seg_ + 2 - Cfg::kSegs;
        const int g2 = (grp_ + 2) % 3;
        static_assert(Cfg::kUnits % 3 == 0, "the unit loop is unrolled by three");
        f32x4 f0h = ACX_W2_RD(base, 0, 0), f0l = ACX_W2_RD(base, 0, 1), f1h = ACX_W2_RD(base, 1, 0), f1l = ACX_W2_RD(base, 1, 1),
              f2h = ACX_W2_RD(base, 2, 0), f2l = ACX_W2_RD(base, 2, 1);
        ACX_FENCE
        if constexpr (HV) { ACX_NANO_RANGE(0, 0, Cfg::kNanoHead) }
        ACX_FENCE
#define ACX_P2_UNIT(i_, ch_, cl_, th_, tl_)                                                                     \
            ACX_FENCE                                                                                           \
            ACX_M16(cl_, gh[0], acc[i_][0])                                                                     \
            if ((i_) % Cfg::kDmaStride == 0 && dma) { ACX_WDMA(dseg, (i_) / Cfg::kDmaStride, g2) }              \
            ACX_NANO_AT(HV, 0, 6 * (i_) + 0)                                                                    \
            if constexpr (NPB == 2) { ACX_M16(cl_, gh[1], acc[i_][1]) }                                         \
            if ((i_) + 3 < Cfg::kUnits) cl_ = ACX_W2_RD(base, (i_) + 3, 1);                                     \
            ACX_NANO_AT(HV, 0, 6 * (i_) + 1)                                                                    \
            ACX_M16(ch_, gl[0], acc[i_][0])                                                                     \
            ACX_NANO_AT(HV, 0, 6 * (i_) + 2)                                                                    \
            if constexpr (NPB == 2) { ACX_M16(ch_, gl[1], acc[i_][1]) }                                         \
            ACX_NANO_AT(HV, 0, 6 * (i_) + 3)                                                                    \
            ACX_M16(ch_, gh[0], acc[i_][0])                                                                     \
            if ((i_) + 1 < Cfg::kUnits) ACX_TOUCH2(th_, tl_)                                                    \
            ACX_NANO_AT(HV, 0, 6 * (i_) + 4)                                                                    \
            if constexpr (NPB == 2) { ACX_M16(ch_, gh[1], acc[i_][1]) }                                         \
            if ((i_) + 3 < Cfg::kUnits) ch_ = ACX_W2_RD(base, (i_) + 3, 0);                                     \
            ACX_NANO_AT(HV, 0, 6 * (i_) + 5)                                                                    \
            ACX_FENCE
#pragma unroll
        for (int i = 0; i < Cfg::kUnits; i += 3) {
            ACX_P2_UNIT(i, f0h, f0l, f1h, f1l)
            ACX_P2_UNIT(i + 1, f1h, f1l, f2h, f2l)
            ACX_P2_UNIT(i + 2, f2h, f2l, f0h, f0l)
        }
#undef ACX_P2_UNIT
        ACX_SEG_END(dma, 2, (kOpt & 1) ? kAhead : 0)
    };

    using Opt0 = std::integral_constant<int, 0>;
    using OptA = std::integral_constant<int, PERS ? 1 : 0>;       // kAhead more in flight at the segment's end
    using OptY = std::integral_constant<int, PERS ? 3 : 0>;       // the same + request the next tile's rows
    static_assert(!PERS || (Cfg::kSegs % 3 == 0 && n >= 4 && Cfg::kPieces + kAhead < 64), "ring phase at a tile boundary; vmcnt is 6 bits");
    if (first) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // segments 0 and 1 landed ...
        __syncthreads();                                  // ... for every wave; b1s visible
    }
    // (PERS, later tiles: segment 0 landed behind the barrier of the previous tile's last segment, segment 1 lands by the end of
    // this tile's segment 0, whose wait leaves the previous tile's stores in flight)
    ACX_WSTAMP(0)                                         // tile prologue: y rows, LayerNorm, first two segments' flight
    // segment 0: phase 1 of chunk 0; the first half of its GELU has nothing to ride on
    phase1(std::false_type{}, 0, 0, 0, OptA{});
    ACX_NANO_RANGE(0, 0, Cfg::kNano)
    // segments 2k-1 (phase 1 of chunk k + second half of GELU(k-1)) and 2k (phase 2 of chunk k-1 + first half of GELU(k));
    // ring slot = segment % 3
    int grp = 1;
    for (int k = 1; k < (PERS ? n - 2 : n - 1); ++k) {
        phase1(std::true_type{}, k, 2 * k - 1, grp, Opt0{});
        grp = grp == 2 ? 0 : grp + 1;
        phase2(std::true_type{}, 2 * k, grp, Opt0{});
        grp = grp == 2 ? 0 : grp + 1;
    }
    if constexpr (PERS) {
        // the next tile's rows are requested at the end of segment 2n - 5, behind its last piece; they must have landed by the end
        // of segment 2n - 3 (the pieces requested during 2n - 4 are younger): two segments, ~2 us
        phase1(std::true_type{}, n - 2, 2 * n - 5, grp, OptY{});
        grp = grp == 2 ? 0 : grp + 1;
        phase2(std::true_type{}, 2 * n - 4, grp, OptA{});
        grp = grp == 2 ? 0 : grp + 1;
    }
    phase1(std::true_type{}, n - 1, 2 * n - 3, grp, Opt0{});
    grp = grp == 2 ? 0 : grp + 1;
    // the activations are dead from here on: their registers take the residual x of the tile, requested two segments
    // (~2 us) before the epilogue needs it
    // (C = 384: all but the last kLateX blocks per pixel block -- with all 48 in flight hipcc spilled three of them, each spill a
    // load + s_waitcnt vmcnt(0) + scratch store, i.e. three serial HBM round trips in the middle of the tile; the late ones are
    // requested at the top of the epilogue, into registers the loop has freed by then)
    constexpr int kLateX = C >= 384 ? 3 : 0;
    constexpr int kEarlyX = C / 16 - kLateX;
    f32x4 xr[2][kEarlyX];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        const float* xp = x + mrow[pb] * C + 4 * g4;
#pragma unroll
        for (int cb = 0; cb < kEarlyX; ++cb) xr[pb][cb] = *reinterpret_cast<const f32x4*>(xp + 16 * cb);
    }
    static_assert(!PERS || 2 * kEarlyX == kAhead, "the wait of segment 2n - 2 counts the x loads");
    phase2(std::true_type{}, 2 * n - 2, grp, OptA{});
    grp = grp == 2 ? 0 : grp + 1;
    if constexpr (NPB == 2) { ACX_NANO_RANGE(1, 0, Cfg::kNano) }      // second half of the last chunk's GELU: no phase-1 segment left to ride on
    ACX_PACK_G()
    phase2(std::false_type{}, 2 * n - 1, grp, Opt0{});
#undef ACX_H8
#undef ACX_FENCE
#undef ACX_W1_RD
#undef ACX_W2_RD
#undef ACX_TOUCH2
#undef ACX_BIAS_INIT
#undef ACX_NANO_RANGE
#undef ACX_NANO_CASE
#undef ACX_NANO_ARGS
#undef ACX_NANO_AT
#undef ACX_PACK_G
#undef ACX_SEG_END

    // ---- epilogue: lane (px = l15 of block pb, g4), block cb: channels 16 cb + 4 g4 .. + 3  ->  x = x + out + b2 ---------
    f32x4 xl[2][kLateX > 0 ? kLateX : 1];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int i = 0; i < kLateX; ++i) xl[pb][i] = *reinterpret_cast<const f32x4*>(x + mrow[pb] * C + 4 * g4 + 16 * (kEarlyX + i));
#define ACX_XR(pb_, cb_) ((cb_) < kEarlyX ? xr[pb_][(cb_) < kEarlyX ? (cb_) : 0] : xl[pb_][(cb_) >= kEarlyX ? (cb_) - kEarlyX : 0])
    // masked: rows beyond M are not stored -- each pixel block's stores then sit behind a branch, and hipcc, which cannot
    // know whether the other block's stores were issued, waits for all but 11 vector-memory operations before each of the second
    // block's (a window of 11 stores in flight).  PERS tiles that lie inside M (all but possibly the last) take the unmasked form.
    auto epilogue = [&](auto masked) __attribute__((always_inline)) {
    constexpr bool kMasked = decltype(masked)::value;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
    if constexpr (LNOUT) {
        // last block of the stage in the full forward: the only reader of the new x is the LayerNorm in front of the
        // downsample conv (convnext.py:230-235): write its S16 operand instead (see mlp_fused_split.hip)
        float sum = 0.f;
#pragma unroll
        for (int cb = 0; cb < C / 16; ++cb) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 16 * cb + 4 * g4);
            const f32x4 v = ACX_XR(pb, cb);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[cb][pb][e] = v[e] + fmaf(acc[cb][pb][e], sinv2, bb[e]);
            sum += (acc[cb][pb][0] + acc[cb][pb][1]) + (acc[cb][pb][2] + acc[cb][pb][3]);
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int cb = 0; cb < C / 16; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float u = acc[cb][pb][e] - mean; d = fmaf(u, u, d); }
        d += __shfl_xor(d, 16);
        d += __shfl_xor(d, 32);
        const float sc = kSplitLnScale / sqrtf(d * (1.0f / C) + 1e-6f);
        // an S16 block (8 channels: [8 hi][8 lo]) is shared by the lanes (g4, g4 ^ 1) of a pixel: after a permlane16 swap the
        // even lane holds all 8 hi halves and the odd lane all 8 lo halves -> one 16-byte store each, 32 contiguous bytes per row
        char* op = ln_out + mrow[pb] * (long long)(C * 4) + (g4 >> 1) * 32 + 16 * (g4 & 1);
#pragma unroll
        for (int cb = 0; cb < C / 16; ++cb) {
            unsigned uhi[2], ulo[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                acx_split_pair((acc[cb][pb][2 * e] - mean) * sc, (acc[cb][pb][2 * e + 1] - mean) * sc, uhi[e], ulo[e]);
                acx_pair_swap16(uhi[e], ulo[e]);
            }
            if (!kMasked || valid[pb]) *reinterpret_cast<uint4*>(op + cb * 64) = uint4{uhi[0], uhi[1], ulo[0], ulo[1]};
        }
    } else if (!kMasked || valid[pb]) {
        float* xp = x + mrow[pb] * C + 4 * g4;
#pragma unroll
        for (int cb = 0; cb < C / 16; ++cb) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + 16 * cb + 4 * g4);
            f32x4 v = ACX_XR(pb, cb);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += fmaf(acc[cb][pb][e], sinv2, bb[e]);
            *reinterpret_cast<f32x4*>(xp + 16 * cb) = v;
        }
    }
    }
    };
    if (PERS && (tile + 1) * kPixT <= M) epilogue(std::false_type{});
    else epilogue(std::true_type{});
#undef ACX_XR
    ACX_WSTAMP(4)
    if (!has_next) break;
    // (the stores above stay in flight: the wait at the end of the next tile's segment 0 allows for them)
    tile += gridDim.x;
    ACX_TILE_ROWS(tile)
    first = false;
    }
#undef ACX_WDMA
#undef ACX_Y_LOAD
#undef ACX_BIAS_STAGE
#undef ACX_TILE_ROWS
    ACX_WSTAMP_FLUSH
}

template <int C, int PT, bool LNOUT, int NPB, bool PERS = false>
static int launch_wide_cfg(const BlockW& w, const float* y, float* x, long long M, void* ln_out, hipStream_t s, int max_wgs = 0) {
    using Cfg = WideCfg<C, PT>;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "weight ring does not fit the LDS");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_wide_kernel<C, PT, LNOUT, NPB, PERS>, kCuLdsBytes));
    constexpr int kPixT = Cfg::kWaves * 16 * NPB;
    long long blocks = (M + kPixT - 1) / kPixT;
    if (PERS && blocks > max_wgs) blocks = max_wgs;      // one workgroup per CU this launch may count on; each walks its tiles
    launch_kernel(&mlp_fused_wide_kernel<C, PT, LNOUT, NPB, PERS>, dim3((unsigned)blocks), dim3(Cfg::kThreads), kCuLdsBytes /* CU-exclusive */, s,
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
    // 64-pixel tiles (one 16-pixel block per wave) when all of them -- of every sub-batch in flight -- find a CU at once: a small
    // batch then spreads over twice the CUs at ~0.65 of the time per tile; beyond that the 128-pixel tile is the efficient one.
    // The arithmetic of a pixel does not depend on the choice (same MFMA order, same GELU): results are bit-identical.
    int cus = 0;
    ACX_TRY(cu_count_of_current_device(&cus));
    const int ways = inflight_ways();
    bool half = (M + 63) / 64 * ways <= cus;
    if (const int f = tuning().wide_npb.load(std::memory_order_relaxed)) half = f == 1;
#define ACX_GO(C_, LN_) (half ? launch_wide_cfg<C_, 1, LN_, 1>(w, y, x, M, LN_ ? ln_out : nullptr, s) : launch_wide_cfg<C_, 1, LN_, 2>(w, y, x, M, LN_ ? ln_out : nullptr, s))
    if (C == 384) return ln_out ? ACX_GO(384, true) : ACX_GO(384, false);
    if (C == 192) {
        // the persistent form once every CU of this launch's share has more than one tile to walk (ACX_WIDE_PERSIST = 0 | 1 forces)
        const int share = cus / ways > 0 ? cus / ways : 1;
        bool pers = !half && (M + 127) / 128 > share;
        if (const int f = tuning().wide_pers.load(std::memory_order_relaxed)) pers = f == 1 && !half;
        if (pers) return ln_out ? launch_wide_cfg<192, 1, true, 2, true>(w, y, x, M, ln_out, s, share)
                                : launch_wide_cfg<192, 1, false, 2, true>(w, y, x, M, nullptr, s, share);
        return ln_out ? ACX_GO(192, true) : ACX_GO(192, false);
    }
#undef ACX_GO
    ACX_FAIL(ACX_ERR_SHAPE, "wide fused MLP: unsupported channel count %d", C);
}

}  // namespace acx
