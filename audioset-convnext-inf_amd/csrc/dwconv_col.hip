// K3c -- depthwise 7x7 convolution (Block.dwconv, convnext.py:58-60 called at :76), column-streaming form (round 4).
//
// The ring / tile kernels of dwconv.hip keep OUTPUTS stationary (2 rows x 7 pixels per thread): every output tile re-reads
// its 8 input rows and the 49 weights from the LDS (153 reads per 686 packed FMAs), the workgroup meets at two barriers per
// tile, and staging goes through registers.  They ran at 0.48 of the HBM roofline, bound by instruction issue at two waves
// per SIMD (profiles/r03_x_split_pmc_per_kernel.csv: valu 0.32, wait_inst 0.36).
//
// Here the INPUT row is what streams and everything else is stationary:
//   * a WAVE owns a column of the stacked batch: 28 output pixels x 32 channels (W >= 28), 14 x 64 (W = 14) or 7 x 128
//     (W = 7) -- always 896 outputs per image row, i.e. 6 such columns cover a row of any stage -- and walks DOWN a
//     segment of the batch stacked as one tall image (dwconv.hip: H rows of a clip, 3 rows of zeros, H rows of the next);
//   * a lane (16 lanes x float2 = a pixel's 32-channel slice = one 128-B line) keeps the 49 weights of its two channels
//     (98 registers) and the accumulators of the 7 output rows x 7 pixels an input row contributes to (98 registers);
//   * each input row is read from the LDS ONCE (13 reads of 8 B per lane) and meets all 49 taps: 343 v_pk_fma_f32 per row
//     and lane, after which one output row is complete, leaves for HBM (7 stores of 8 B: 128-B lines) and its registers
//     start the row seven further down.  No weight re-reads, no halo rows (6 extra rows per SEGMENT instead of 6 per
//     8-row tile), 26 FMAs per LDS read instead of 4.5;
//   * rows arrive by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a ring of 7 rows that is PRIVATE to the
//     wave: no barrier anywhere in the kernel, a counted s_waitcnt vmcnt is the only synchronisation.  kD rows (22 KB per
//     wave, 88 KB per CU) are in flight while a row is multiplied.  The ring has as many rows as the register rotation
//     has phases (7), so that slot addresses are immediates of the 7-fold unrolled loop;
//   * out-of-image columns are LDS slots zeroed once and never written (the DMA lanes that would fill them are masked
//     off); rows outside the image (the 3 rows between clips) skip their FMAs.
// The accumulation order of an output element is the one of dwconv.hip -- bias, then kernel rows top to bottom, each left
// to right, one fused multiply-add per tap -- so both forms give the same bits and the launcher may pick by launch size.
//
// One wave per SIMD (230 registers: a second wave would not fit, and would double the segments and their halo rows).
// A lone wave issues v_pk_fma_f32 at ~4.6 cycles (peak 4); what it cannot hide behind a partner it hides behind its own
// deep prefetch: the FMAs (~85 us at B = 64 in stage 0) run entirely under the memory time (~125 us).
#include <type_traits>

#include "acx_internal.h"

namespace acx {

typedef float dwc_f32x2 __attribute__((ext_vector_type(2)));
typedef float dwc_f32x4 __attribute__((ext_vector_type(4)));

// R: rows of the wave's ring.  7 = the phases of the accumulator rotation: slot addresses are immediates of the unrolled loop
// (one wave per SIMD has the LDS for it).  Smaller rings -- two waves per SIMD, 20 KB of LDS each -- walk their slots with two
// running byte offsets.
template <int W, bool BF, int R = 7>
struct DwColCfg {
    static constexpr int kC = 96 * 56 / W;                        // channels of the stage with this width
    static constexpr int kStrips = W >= 28 ? 4 : W / 7;           // strips of 7 output pixels per wave (16 lanes each)
    static constexpr int kSlices = 4 / kStrips;                   // 32-channel slices per wave
    static constexpr int kPx = 7 * kStrips;                       // output pixels per image row and (wave, slice)
    static constexpr int kHalves = W / kPx;                       // waves across the image width (2 for W = 56)
    static constexpr int kUnits = kHalves * (kC / 32 / kSlices);  // wave columns per image row: 6 in every stage
    static constexpr int kHalo = W == 7 ? 0 : 3;                  // W = 7: the halo columns are all outside the image: their taps are skipped
    static constexpr int kSlots = kPx + 2 * kHalo;                // pixel slots of a ring row per slice
    static constexpr int kEsz = BF ? 2 : 4;
    static constexpr int kSlotB = 32 * kEsz;                      // bytes of a pixel's 32-channel slice
    static constexpr int kLanesPerSlot = kSlotB / 16;
    static constexpr int kSPP = 1024 / kSlotB;                    // slots per 1-KB DMA piece
    static constexpr int kReal = kHalves == 2 ? kPx + 3 : W;      // in-image columns a wave reads (W = 56: 31)
    static constexpr int kPiecesPerSlice = (kReal + kSPP - 1) / kSPP;
    static constexpr int kPieces = kSlices * kPiecesPerSlice;
    static constexpr int kRowB = kSlices * kSlots * kSlotB;       // bytes of a ring row
    static constexpr int kRing = R;
    static constexpr int kD = (BF ? 6 : 5) < R - 1 ? (BF ? 6 : 5) : R - 1;   // rows in flight ahead of the one being multiplied
    // Every step issues exactly kPieces DMAs and 7 stores (invalid ones go to a sink), in that order behind its wait:
    // the row of step t was requested at step t - kD, followed by that step's 7 stores and kD - 1 whole steps.
    static constexpr int kWait = 7 + (kD - 1) * (kPieces + 7);
    static constexpr int kWaveLds = kRing * kRowB;
    static constexpr size_t kLdsBytes = (size_t)4 * kWaveLds;
    static constexpr int kGRowB = W * kC * kEsz;                  // bytes of an image row in HBM
    static_assert(kWait <= 63, "vmcnt is a 6-bit counter");
    static_assert(kD < kRing, "the row being read must not be a DMA target");
    static_assert(kLdsBytes <= 160 * 1024, "ring does not fit the LDS");
    static_assert(kUnits == 6, "six wave columns per image row");
};

// lab builds only (tools/lab/dwcol_lab.hip): ACX_DWC_ABLATE = 1 one FMA in seven, 2 no DMA, 3 no stores (timing, wrong results)
#ifndef ACX_DWC_ABLATE
#define ACX_DWC_ABLATE 0
#endif
// Accumulators are tied in/out operands ("+v") in BOTH forms: with a fresh output ("=v") inside the conditional kernel-row
// blocks every accumulator becomes a phi of two values and hipcc shuffles them through AGPRs (851 v_accvgpr per loop body).
// lab builds only: -DACX_DWC_STAMPS -> s_memtime at six points of every wave, written to acx_dwc_stamps[item][8]
#ifdef ACX_DWC_STAMPS
__device__ unsigned long long acx_dwc_stamps[4096 * 8];
#define ACX_DWC_STAMP(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0 && item < 4096) acx_dwc_stamps[item * 8 + (k_)] = t_; }
#else
#define ACX_DWC_STAMP(k_)
#endif
#define ACX_DWC_PKFMA(acc_, a_, b_) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_))
#define ACX_DWC_PKFMA_INIT(acc_, a_, b_, c_) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(acc_) : "v"(a_), "v"(b_), "v"(c_))

template <int W, bool BF>
struct DwColState {
    dwc_f32x2 acc[7][7];      // [phase slot][pixel]
    dwc_f32x2 wt[49];
    dwc_f32x2 bias;
};

// lanes of piece k (within a slice) that fetch in-image columns
template <int W, bool BF, int K>
constexpr unsigned long long dwc_mask() {
    using Cfg = DwColCfg<W, BF>;
    unsigned long long m = 0;
    for (int lane = 0; lane < 64; ++lane)
        if (K * Cfg::kSPP + lane / Cfg::kLanesPerSlot < Cfg::kReal) m |= 1ull << lane;
    return m;
}

// A row's pieces in ONE statement with ONE M0 write: the immediate offset of global_load_lds moves the LDS destination AND the
// source by the same amount (split_math.h, acx_glds16_run), so piece p is issued with its LDS distance from piece 0 as the
// immediate, and its lane offsets are stored with that distance taken out again (+ kDwcAdj, which the base pointer gives back:
// the offsets stay non-negative).  Full pieces first, then EXEC is narrowed once for the partial ones (their lanes beyond the
// image stay out: those LDS slots hold zeros or the next slice).  6-8 instructions per row instead of 14.
constexpr int kDwcAdj = 4096;
template <int W, bool BF, int P>
constexpr int dwc_piece_lds_off() {          // LDS distance of piece P from piece 0
    using Cfg = DwColCfg<W, BF>;
    return (P / Cfg::kPiecesPerSlice) * Cfg::kSlots * Cfg::kSlotB + (P % Cfg::kPiecesPerSlice) * 1024;
}
template <int W, bool BF>
__device__ __forceinline__ void dwc_issue_row(const char* src, const unsigned (&voff)[DwColCfg<W, BF>::kPieces], unsigned lds0) {
    using Cfg = DwColCfg<W, BF>;
    if (ACX_DWC_ABLATE == 2) return;
    const char* base = src - kDwcAdj;
    constexpr int kLast = Cfg::kPiecesPerSlice - 1;
    constexpr unsigned long long kMask = dwc_mask<W, BF, kLast>();          // the partial piece(s): the last of every slice
    static_assert(kMask != ~0ull, "every layout ends its slices with a partial piece");
    unsigned long long keep;
    if constexpr (Cfg::kPieces == 4 && Cfg::kSlices == 1) {                 // W = 56 / 28 fp32: 0 1 2 full, 3 partial
        asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, exec\n\t"
                     "global_load_lds_dwordx4 %2, %6 offset:%c7\n\tglobal_load_lds_dwordx4 %3, %6 offset:%c8\n\tglobal_load_lds_dwordx4 %4, %6 offset:%c9\n\t"
                     "s_mov_b64 exec, %11\n\tglobal_load_lds_dwordx4 %5, %6 offset:%c10\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "s"(lds0), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(base),
                       "n"(dwc_piece_lds_off<W, BF, 0>()), "n"(dwc_piece_lds_off<W, BF, 1>()), "n"(dwc_piece_lds_off<W, BF, 2>()), "n"(dwc_piece_lds_off<W, BF, 3>()),
                       "s"(kMask) : "memory");
    } else if constexpr (Cfg::kPieces == 4 && Cfg::kSlices == 2) {          // W = 14 fp32: 0, 2 full; 1, 3 partial
        asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, exec\n\t"
                     "global_load_lds_dwordx4 %2, %6 offset:%c7\n\tglobal_load_lds_dwordx4 %4, %6 offset:%c9\n\t"
                     "s_mov_b64 exec, %11\n\tglobal_load_lds_dwordx4 %3, %6 offset:%c8\n\tglobal_load_lds_dwordx4 %5, %6 offset:%c10\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "s"(lds0), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(base),
                       "n"(dwc_piece_lds_off<W, BF, 0>()), "n"(dwc_piece_lds_off<W, BF, 1>()), "n"(dwc_piece_lds_off<W, BF, 2>()), "n"(dwc_piece_lds_off<W, BF, 3>()),
                       "s"(kMask) : "memory");
    } else if constexpr (Cfg::kPieces == 4) {                               // W = 7: four slices, one partial piece each
        asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, %11\n\t"
                     "global_load_lds_dwordx4 %2, %6 offset:%c7\n\tglobal_load_lds_dwordx4 %3, %6 offset:%c8\n\t"
                     "global_load_lds_dwordx4 %4, %6 offset:%c9\n\tglobal_load_lds_dwordx4 %5, %6 offset:%c10\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "s"(lds0), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(base),
                       "n"(dwc_piece_lds_off<W, BF, 0>()), "n"(dwc_piece_lds_off<W, BF, 1>()), "n"(dwc_piece_lds_off<W, BF, 2>()), "n"(dwc_piece_lds_off<W, BF, 3>()),
                       "s"(kMask) : "memory");
    } else if constexpr (Cfg::kPieces == 2 && Cfg::kSlices == 1) {          // W = 56 / 28 bf16: 0 full, 1 partial
        asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, exec\n\t"
                     "global_load_lds_dwordx4 %2, %4 offset:%c5\n\t"
                     "s_mov_b64 exec, %7\n\tglobal_load_lds_dwordx4 %3, %4 offset:%c6\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "s"(lds0), "v"(voff[0]), "v"(voff[1]), "s"(base),
                       "n"(dwc_piece_lds_off<W, BF, 0>()), "n"(dwc_piece_lds_off<W, BF, 1>()), "s"(kMask) : "memory");
    } else {                                                                // W = 14 bf16: two slices, one partial piece each
        static_assert(Cfg::kPieces == 2 && Cfg::kSlices == 2, "piece layout");
        asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, %7\n\t"
                     "global_load_lds_dwordx4 %2, %4 offset:%c5\n\tglobal_load_lds_dwordx4 %3, %4 offset:%c6\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "s"(lds0), "v"(voff[0]), "v"(voff[1]), "s"(base),
                       "n"(dwc_piece_lds_off<W, BF, 0>()), "n"(dwc_piece_lds_off<W, BF, 1>()), "s"(kMask) : "memory");
    }
}

// One input row: phase I of the rotation (static), ring slot I, kernel rows KLO .. KHI (static).
//   rowp    LDS address of this lane's first input column in the row's ring slot
//   real    the row is inside an image (wave-uniform): otherwise it contributes nothing
// A segment's first six input rows only meet the kernel rows whose output row lies inside the segment (KHI = 0 .. 5), its
// last six likewise (KLO = 1 .. 6): the halo rows of a segment cost their loads, not their FMAs -- and the kernel-row range is
// a template argument, the FMA stream stays straight-line code (a run-time test per kernel row made hipcc route weights and
// accumulators through AGPRs: 730 v_accvgpr + 500 v_mov per loop body).
// Order: input column by input column, left to right; a column meets every kernel row before the next one is touched (each
// column is first needed 7+ FMAs after the one before it: the wave waits for the first LDS read only; an accumulator returns
// after >= 7 other FMAs).  For one output element that is kernel column 0 .. 6 -- the order of dwconv.hip.
template <int W, bool BF, int I, int KLO, int KHI>
__device__ __forceinline__ void dwc_row(DwColState<W, BF>& st, const char* rowp, bool real) {
    using Cfg = DwColCfg<W, BF>;
    constexpr int slot0 = (I + 6) % 7;            // the output row that starts here (kernel row 0)
    if (real) {
        dwc_f32x2 in[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            if (W == 7 && (j < 3 || j > 9)) continue;           // columns outside the image: zero, taps skipped
            const int s = W == 7 ? j - 3 : j;
            if constexpr (BF) {
                const unsigned u = *reinterpret_cast<const unsigned*>(rowp + s * Cfg::kSlotB);
                in[j] = dwc_f32x2{acx_bf16_lo(u), acx_bf16_hi(u)};
            } else {
                in[j] = *reinterpret_cast<const dwc_f32x2*>(rowp + s * Cfg::kSlotB);
            }
        }
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            if (W == 7 && (j < 3 || j > 9)) continue;
#pragma unroll
            for (int ky = KLO; ky <= KHI; ++ky) {
                const int a = (I - ky + 6 + 7) % 7;
#pragma unroll
                for (int kx = (j > 6 ? j - 6 : 0); kx <= (j > 6 ? 6 : j); ++kx) {
                    const int p = j - kx;
                    if (ACX_DWC_ABLATE == 1 && kx > 0) continue;
                    // the first tap of a fresh output row adds to the bias; W = 7: its first in-image tap is kx = 3 - p
                    const bool first = ky == 0 && (W == 7 ? (kx == (p < 3 ? 3 - p : 0)) : kx == 0);
                    if (first) { ACX_DWC_PKFMA_INIT(st.acc[a][p], in[j], st.wt[ky * 7 + kx], st.bias); }
                    else { ACX_DWC_PKFMA(st.acc[a][p], in[j], st.wt[ky * 7 + kx]); }
                }
            }
        }
    } else if (KLO == 0) {
#pragma unroll
        for (int p = 0; p < 7; ++p) st.acc[slot0][p] = st.bias;
    }
}

// S0: a segment of n_out = 7 k7 - S0 output rows (any length >= 7) runs as a virtual segment of 7 k7 rows that starts S0 rows
// higher and whose first S0 steps are not executed: the first and the last six executed steps then still sit at fixed phases of
// the rotation, for every segment length (a multiple of 7 only would leave up to 12 % of the wave slots of stage 2 empty).
template <int W, bool BF, int R, int WPS, int S0>
__global__ __launch_bounds__(256, WPS) void dwconv7_col_kernel(const void* __restrict__ x_, void* __restrict__ y_,
                                                             const float* __restrict__ wt /*[49][C]*/,
                                                             const float* __restrict__ bias, void* __restrict__ sink_,
                                                             int B, int H, int k7 /* output rows per segment / 7 */, int n_items,
                                                             unsigned magic /* floor(2^32 / (H + 3)) + 1 */) {
    using Cfg = DwColCfg<W, BF, R>;
    static_assert(Cfg::kLdsBytes * WPS <= 160 * 1024, "the rings of a CU's waves do not fit the LDS");
    constexpr int C = Cfg::kC, kEsz = Cfg::kEsz, D = Cfg::kD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (a wave past the last item repeats it -- identical stores: no early exit, so that every kernel argument is loaded at once)
    const int item = min((int)blockIdx.x * 4 + wave, n_items - 1);
    ACX_DWC_STAMP(0)
    const int unit = item % Cfg::kUnits, seg = item / Cfg::kUnits;
    const int half = unit % Cfg::kHalves, sg = unit / Cfg::kHalves;
    const int cbase = sg * Cfg::kSlices * 32;
    const int first_real = (Cfg::kHalves == 2 && half == 1) ? 0 : Cfg::kHalo;
    const int col_base = (Cfg::kHalves == 2 && half == 1) ? Cfg::kPx - 3 : -Cfg::kHalo;   // column of slot 0

    char* const ring = smem + wave * Cfg::kWaveLds;
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;

    // ---- compute roles: 16 lanes x float2 = one pixel's 32-channel slice ----
    const int g4 = lane >> 4, l16 = lane & 15;
    const int strip = g4 % Cfg::kStrips, sl_i = g4 / Cfg::kStrips;
    const int ch = cbase + sl_i * 32 + 2 * l16;
    DwColState<W, BF> st;
    // The 49 + 1 weight loads go out first -- 220 cycles of issue; their round trip then runs under the index arithmetic, the halo
    // zeroing and the issue of the first rows (nothing below touches them before the rows have been requested) ...
#pragma unroll
    for (int t = 0; t < 49; ++t) st.wt[t] = *reinterpret_cast<const dwc_f32x2*>(wt + t * C + ch);
    st.bias = *reinterpret_cast<const dwc_f32x2*>(bias + ch);
    const char* const rd = ring + sl_i * Cfg::kSlots * Cfg::kSlotB + strip * 7 * Cfg::kSlotB + l16 * (BF ? 4 : 8);
    const unsigned yoff = (unsigned)(((half * Cfg::kPx + strip * 7) * C + ch) * kEsz);

    // ---- DMA roles: a 1-KB piece = kSPP pixel slots, kLanesPerSlot lanes x 16 B each ----
    unsigned voff[Cfg::kPieces];
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) {
        const int sl = p / Cfg::kPiecesPerSlice, k = p % Cfg::kPiecesPerSlice;
        int col = col_base + first_real + k * Cfg::kSPP + lane / Cfg::kLanesPerSlot;
        col = col < 0 ? 0 : (col >= W ? W - 1 : col);                 // masked lanes: any valid address
        voff[p] = (unsigned)((col * C + cbase + sl * 32) * kEsz + (lane % Cfg::kLanesPerSlot) * 16) +
                  (unsigned)(kDwcAdj - (sl * Cfg::kSlots * Cfg::kSlotB + k * 1024));          // see dwc_issue_row
    }

    // ---- the segment: 7 k7 output rows [vb, ve) of the stacked image (stacked row v = clip * (H + 3) + row; row >= H: one of
    // the three zero rows between clips; the last segment may reach past the last clip), input rows [vb - 3, ve + 3).  Step u
    // multiplies input row vb - 3 + u and completes output row vb - 6 + u; the 7 k7 + 6 steps run as groups of seven.
    const int Hp = H + 3;
    const int n_out = 7 * k7 - S0;
    const int vb = seg * n_out;
    const int vbv = vb - S0;                                        // first output row of the virtual segment
    const char* const x0 = reinterpret_cast<const char*>(x_);
    const char* pf_ptr;                                             // next in-image row at or after the prefetch cursor
    {
        const int v0 = vb - 3 + Hp;                                 // >= 0
        const int n1 = (int)__umulhi((unsigned)v0, magic), r0 = v0 - n1 * Hp;
        pf_ptr = x0 + ((long long)(n1 - 1) * H + (r0 < H ? r0 : H)) * Cfg::kGRowB;
    }
    char* out_ptr;                                                  // next in-image output row at or after vb
    {
        const int n = (int)__umulhi((unsigned)vb, magic), r = vb - n * Hp;
        out_ptr = reinterpret_cast<char*>(y_) + ((long long)n * H + (r < H ? r : H)) * Cfg::kGRowB;
    }
    // Stores of rows that are not this wave's (the six steps above its segment, the rows between clips) go to a sink.  Every wave
    // gets a window of its own there (modulo kDwSinkWindows): a thousand waves storing to the same lines serialise in one L2 channel.
    char* const sink = reinterpret_cast<char*>(sink_) + (size_t)(item % kDwSinkWindows) * kDwSinkWindowBytes;

    // Row flags of a group of seven steps, computed by the lanes and read back as wave-uniform bit masks (one v_cmp each):
    // lane l looks at step u = 7 g - 3 + l; bit l of rmask: that input row is inside an image; bit l of omask: it is an
    // output row this wave stores.
    unsigned rmask = 0, omask = 0;
#define ACX_DWC_FLAGS(g_)                                                                                       \
    {                                                                                                           \
        const int u_ = 7 * (g_) - 3 + lane;                           /* virtual step */                        \
        const int v_ = vbv - 3 + u_;                                                                            \
        const unsigned vv_ = (unsigned)(v_ + 9 * Hp);                 /* >= 0: v_ >= -15, Hp >= 4 */            \
        const unsigned n_ = __umulhi(vv_, magic);                                                               \
        const unsigned r_ = vv_ - n_ * (unsigned)Hp;                                                            \
        const bool real_ = v_ >= 0 && r_ < (unsigned)H && n_ < (unsigned)(B + 9) && u_ >= S0 && u_ < 7 * k7 + 6; \
        rmask = (unsigned)__ballot(real_);                                                                      \
        omask = (unsigned)__ballot(real_ && u_ >= S0 + 3 && u_ < 7 * k7 + 3);                                   \
    }

    // The out-of-image column slots of the ring are zeroed once and never written again (the three slots left and right of every
    // slice's pixels; W = 56: only the outer side of each half), then the first kD rows are requested -- BEFORE the weights: the
    // prologue is one memory latency, not two (weights first cost 7 k cycles per wave before the first row was even requested).
    if constexpr (Cfg::kHalo > 0) {
        constexpr int kChunks = Cfg::kSlotB / 16;                              // 16-byte chunks per slot
        constexpr int kZ = R * Cfg::kSlices * 2 * Cfg::kHalo * kChunks;      // chunks to zero
        for (int i = lane; i < kZ; i += 64) {
            const int c = i % kChunks, s6 = (i / kChunks) % (2 * Cfg::kHalo), rs = i / kChunks / (2 * Cfg::kHalo);
            const int slot = s6 < Cfg::kHalo ? s6 : Cfg::kPx + s6;              // 0, 1, 2, kPx + 3, kPx + 4, kPx + 5
            *reinterpret_cast<dwc_f32x4*>(ring + (rs / Cfg::kSlices) * Cfg::kRowB + ((rs % Cfg::kSlices) * Cfg::kSlots + slot) * Cfg::kSlotB + 16 * c) =
                dwc_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // row u of the current group's window is bit u - 7 g + 3
    // Every step requests a row (the counted wait relies on it); a step whose row is not part of an image re-requests the
    // wave's latest real row (from the cache) -- never one address for all waves: that serialises in one L2 channel.
    const char* safe_src = pf_ptr;
    unsigned cur_off = 0, pf_off = (unsigned)(D * Cfg::kRowB);      // R < 7: ring offsets of the row being read / requested
#define ACX_DWC_PREFETCH(bit_, off_)                                                                            \
    {                                                                                                           \
        const bool real_ = (rmask >> (bit_)) & 1u;                                                              \
        const char* src_ = real_ ? pf_ptr : safe_src;                                                           \
        safe_src = src_;                                                                                        \
        pf_ptr += real_ ? Cfg::kGRowB : 0;                                                                      \
        dwc_issue_row<W, BF>(src_, voff, ring_lds + (off_) + (unsigned)(first_real * Cfg::kSlotB));             \
    }
    ACX_DWC_FLAGS(0)
#pragma unroll
    for (int i = 0; i < D; ++i) ACX_DWC_PREFETCH(S0 + i + 3, (unsigned)((R == 7 ? (S0 + i) % 7 : i) * Cfg::kRowB))
    ACX_DWC_STAMP(1)
    // ... and are first used here
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int p = 0; p < 7; ++p) st.acc[a][p] = st.bias;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // once: the steady state below counts whole steps
    ACX_DWC_STAMP(2)

#define ACX_DWC_STEP(I_, KLO_, KHI_)                                                                            \
    {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ACX_DWC_ABLATE >= 2 ? 0 : Cfg::kWait) : "memory");           \
        ACX_DWC_PREFETCH((I_) + D + 3, (R == 7 ? (unsigned)((((I_) + D) % 7) * Cfg::kRowB) : pf_off))           \
        dwc_row<W, BF, I_, KLO_, KHI_>(st, rd + (R == 7 ? (unsigned)((I_) * Cfg::kRowB) : cur_off), (rmask >> ((I_) + 3)) & 1u); \
        if (R != 7) {                                                                                           \
            cur_off = cur_off + Cfg::kRowB == (unsigned)Cfg::kWaveLds ? 0u : cur_off + Cfg::kRowB;              \
            pf_off = pf_off + Cfg::kRowB == (unsigned)Cfg::kWaveLds ? 0u : pf_off + Cfg::kRowB;                 \
        }                                                                                                       \
        const bool out_ok_ = (omask >> (I_)) & 1u;                                                              \
        char* dst_ = (out_ok_ ? out_ptr : sink) + yoff;                                                         \
        out_ptr += out_ok_ ? Cfg::kGRowB : 0;                                                                   \
        _Pragma("unroll") for (int p = 0; p < (ACX_DWC_ABLATE == 3 ? 0 : 7); ++p) {                             \
            if constexpr (BF) *reinterpret_cast<unsigned*>(dst_ + p * C * kEsz) = acx_pack_bf16x2(st.acc[I_][p].x, st.acc[I_][p].y); \
            else *reinterpret_cast<dwc_f32x2*>(dst_ + p * C * kEsz) = st.acc[I_][p];                            \
        }                                                                                                       \
    }
    // The first six executed steps (virtual steps S0 .. S0 + 5) meet kernel rows 0 .. step - S0 only (the output rows above the
    // segment are not this wave's); they sit in group 0 and, for S0 >= 2, in the head of group 1.
#define ACX_DWC_STEP_G0(I_) if constexpr ((I_) >= S0) ACX_DWC_STEP(I_, 0, ((I_) - S0 < 6 ? (I_) - S0 : 6))
    ACX_DWC_STEP_G0(0) ACX_DWC_STEP_G0(1) ACX_DWC_STEP_G0(2) ACX_DWC_STEP_G0(3) ACX_DWC_STEP_G0(4) ACX_DWC_STEP_G0(5) ACX_DWC_STEP_G0(6)
#undef ACX_DWC_STEP_G0
    if constexpr (S0 >= 2) {
        ACX_DWC_FLAGS(1)
#define ACX_DWC_STEP_G1(I_) ACX_DWC_STEP(I_, 0, ((I_) + 7 - S0 < 6 ? (I_) + 7 - S0 : 6))
        ACX_DWC_STEP_G1(0) ACX_DWC_STEP_G1(1) ACX_DWC_STEP_G1(2) ACX_DWC_STEP_G1(3) ACX_DWC_STEP_G1(4) ACX_DWC_STEP_G1(5) ACX_DWC_STEP_G1(6)
#undef ACX_DWC_STEP_G1
    }
    ACX_DWC_STAMP(3)
    for (int g = S0 >= 2 ? 2 : 1; g < k7; ++g) {
        ACX_DWC_FLAGS(g)
        ACX_DWC_STEP(0, 0, 6) ACX_DWC_STEP(1, 0, 6) ACX_DWC_STEP(2, 0, 6) ACX_DWC_STEP(3, 0, 6) ACX_DWC_STEP(4, 0, 6) ACX_DWC_STEP(5, 0, 6) ACX_DWC_STEP(6, 0, 6)
    }
    // the last six input rows: kernel rows step + 1 .. 6 (the output rows below the segment belong to the next wave)
    ACX_DWC_STAMP(4)
    ACX_DWC_FLAGS(k7)
    ACX_DWC_STEP(0, 1, 6) ACX_DWC_STEP(1, 2, 6) ACX_DWC_STEP(2, 3, 6) ACX_DWC_STEP(3, 4, 6) ACX_DWC_STEP(4, 5, 6) ACX_DWC_STEP(5, 6, 6)
    ACX_DWC_STAMP(5)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ACX_DWC_STAMP(6)
#undef ACX_DWC_STEP
#undef ACX_DWC_PREFETCH
#undef ACX_DWC_FLAGS
}

// target_waves: how many waves the launch should consist of (one per SIMD of the CUs it may use).  Every wave takes a
// segment of 7 k output rows of the stacked batch (the rotation has seven phases: the first and the last six steps then sit at
// fixed phases and are compiled with their reduced kernel-row ranges).
template <int W, bool BF, int R, int WPS, int S0>
static int launch_dw_col_s0(const void* x, void* y, const float* wt, const float* bias, void* sink, int B, int H, int k7, int n_items, hipStream_t s) {
    using Cfg = DwColCfg<W, BF, R>;
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &dwconv7_col_kernel<W, BF, R, WPS, S0>, Cfg::kLdsBytes));
    launch_kernel(&dwconv7_col_kernel<W, BF, R, WPS, S0>, dim3((unsigned)((n_items + 3) / 4)), dim3(256), Cfg::kLdsBytes, s,
        x, y, wt, bias, sink, B, H, k7, n_items, (unsigned)(0x100000000ull / (unsigned)(H + 3)) + 1u);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// target_waves: how many waves the launch should consist of (one per SIMD of the CUs it may use).  Every wave takes a segment
// of n_out = ceil(rows / segments) output rows of the stacked batch; the kernel instantiation for n_out's remainder modulo the
// rotation's seven phases (S0) runs it.
template <int W, bool BF, int R, int WPS>
static int launch_dw_col_cfg(const void* x, void* y, const float* wt, const float* bias, void* sink, int B, int H,
                             int target_waves, hipStream_t s) {
    using Cfg = DwColCfg<W, BF, R>;
    const long long Vt = (long long)B * (H + 3) - 3;
    // exactness of v / (H + 3) by multiply-high needs (stacked rows + 9 (H + 3) + slack) * (H + 3) < 2^32
    if ((Vt + 16ll * (H + 3) + 64) * (H + 3) >= 0xffffffffll)
        ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: batch too tall for one launch (%d clips of %d rows)", B, H);
    long long segs = (long long)target_waves * WPS / Cfg::kUnits;
    if (segs < 1) segs = 1;
    long long n_out = (Vt + segs - 1) / segs;                          // output rows per segment
    const long long nmin = WPS == 2 ? 7 : kDwColMinRows;
    if (n_out < nmin) n_out = nmin;
    const int s0 = (int)((7 - n_out % 7) % 7);
    const int k7 = (int)((n_out + s0) / 7);
    const long long n_seg = (Vt + n_out - 1) / n_out;
    const int n_items = (int)(n_seg * Cfg::kUnits);
    switch (s0) {
        case 0: return launch_dw_col_s0<W, BF, R, WPS, 0>(x, y, wt, bias, sink, B, H, k7, n_items, s);
        case 1: return launch_dw_col_s0<W, BF, R, WPS, 1>(x, y, wt, bias, sink, B, H, k7, n_items, s);
        case 2: return launch_dw_col_s0<W, BF, R, WPS, 2>(x, y, wt, bias, sink, B, H, k7, n_items, s);
        case 3: return launch_dw_col_s0<W, BF, R, WPS, 3>(x, y, wt, bias, sink, B, H, k7, n_items, s);
        case 4: return launch_dw_col_s0<W, BF, R, WPS, 4>(x, y, wt, bias, sink, B, H, k7, n_items, s);
        case 5: return launch_dw_col_s0<W, BF, R, WPS, 5>(x, y, wt, bias, sink, B, H, k7, n_items, s);
        default: return launch_dw_col_s0<W, BF, R, WPS, 6>(x, y, wt, bias, sink, B, H, k7, n_items, s);
    }
}

// Occupancy by stage: the HBM-bound stages 0-1 in fp32 run one wave per SIMD (long segments: few halo rows, a deep ring);
// wherever the FMAs are what takes the time -- the cache-resident stages 2-3, and every stage once the activations are bf16 --
// two waves per SIMD (a second wave fills the issue slots a lone wave leaves: v_pk_fma_f32 issues every 5 cycles from one
// wave, every 4 from two) on half-length segments, whose extra halo rows come from the cache or weigh half.
int launch_dwconv_col(const void* x, void* y, const float* wt, const float* bias, void* sink, int B, int H, int W,
                      bool act_bf16, int target_waves, hipStream_t s) {
    // bf16 activations never come here in production: every launch of `bf16a` takes the matrix-pipe kernel (dwconv_mfma.hip), and
    // the diagnostic switch ACX_DW_MFMA=0 falls back to the tile / ring kernels of dwconv.hip (same bits as this kernel).  Round 6
    // removed the 21 bf16 instantiations of this file (1.2 MB of code objects no launch selected).
    if (act_bf16) ACX_FAIL(ACX_ERR_STATE, "dwconv7 (column form): bf16 activations take the matrix-pipe kernel");
    switch (W) {
        case 56: return launch_dw_col_cfg<56, false, 7, 1>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 28: return launch_dw_col_cfg<28, false, 7, 1>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 14: return launch_dw_col_cfg<14, false, 4, 2>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 7: return launch_dw_col_cfg<7, false, 5, 2>(x, y, wt, bias, sink, B, H, target_waves, s);
        default: ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: unsupported width %d (expected 56/28/14/7)", W);
    }
}

}  // namespace acx
