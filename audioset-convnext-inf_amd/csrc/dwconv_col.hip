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

template <int W, bool BF>
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
    static constexpr int kRing = 7;                               // = phases of the accumulator rotation
    static constexpr int kD = BF ? 6 : 5;                         // rows in flight ahead of the one being multiplied
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
#define ACX_DWC_PKFMA(d_, a_, b_, c_) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d_) : "v"(a_), "v"(b_), "v"(c_))

// one 1-KB piece: lanes of `mask` fetch 16 bytes each from base + voff into lds + 16 lane
template <unsigned long long MASK>
__device__ __forceinline__ void dwc_piece(const char* base, unsigned voff, unsigned lds) {
    if constexpr (MASK == ~0ull) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds), "v"(voff), "s"(base) : "memory");
    } else {
        unsigned long long keep;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %4\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "s"(lds), "v"(voff), "s"(base), "s"(MASK) : "memory");
    }
}

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

template <int W, bool BF, int P>
__device__ __forceinline__ void dwc_issue_pieces(const char* src, const unsigned (&voff)[DwColCfg<W, BF>::kPieces], unsigned lds_row, int first_real) {
    using Cfg = DwColCfg<W, BF>;
    if constexpr (P < Cfg::kPieces) {
        constexpr int sl = P / Cfg::kPiecesPerSlice, k = P % Cfg::kPiecesPerSlice;
        const unsigned lds = lds_row + (unsigned)(sl * Cfg::kSlots * Cfg::kSlotB + k * 1024) + (unsigned)first_real * Cfg::kSlotB;
        if (ACX_DWC_ABLATE != 2) dwc_piece<dwc_mask<W, BF, k>()>(src, voff[P], lds);
        dwc_issue_pieces<W, BF, P + 1>(src, voff, lds_row, first_real);
    }
}

// One input row: phase I of the rotation (static), ring slot I.
//   rd      LDS address of this lane's first input column in ring slot 0
//   real    the row is inside an image (wave-uniform): otherwise it contributes nothing
template <int W, bool BF, int I>
__device__ __forceinline__ void dwc_row(DwColState<W, BF>& st, const char* rd, bool real) {
    using Cfg = DwColCfg<W, BF>;
    constexpr int slot0 = (I + 6) % 7;            // the output row that starts here (kernel row 0)
    if (real) {
        dwc_f32x2 in[13];
        const char* rowp = rd + I * Cfg::kRowB;
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
        // kx outer: an accumulator returns after >= 7 other FMAs, and input column j is first needed in round max(0, j - 6)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int p = 0; p < 7; ++p) {
                    if (W == 7 && (p + kx < 3 || p + kx > 9)) continue;
                    if (ACX_DWC_ABLATE == 1 && kx > 0) continue;
                    const int a = (I - ky + 6 + 7) % 7;
                    // the first tap of a fresh output row adds to the bias; W = 7: its first in-image tap is kx = 3 - p
                    const bool first = ky == 0 && (W == 7 ? (kx == (p < 3 ? 3 - p : 0)) : kx == 0);
                    if (first) { ACX_DWC_PKFMA(st.acc[a][p], in[p + kx], st.wt[ky * 7 + kx], st.bias); }
                    else { ACX_DWC_PKFMA(st.acc[a][p], in[p + kx], st.wt[ky * 7 + kx], st.acc[a][p]); }
                }
    } else {
#pragma unroll
        for (int p = 0; p < 7; ++p) st.acc[slot0][p] = st.bias;
    }
}

template <int W, bool BF>
__global__ __launch_bounds__(256, 1) void dwconv7_col_kernel(const void* __restrict__ x_, void* __restrict__ y_,
                                                             const float* __restrict__ wt /*[49][C]*/,
                                                             const float* __restrict__ bias, void* __restrict__ sink_,
                                                             int B, int H, int n_seg, int n_items) {
    using Cfg = DwColCfg<W, BF>;
    constexpr int C = Cfg::kC, kEsz = Cfg::kEsz;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = (int)blockIdx.x * 4 + wave;
    if (item >= n_items) return;                      // (no barrier in this kernel: a wave may leave alone)
    const int unit = item % Cfg::kUnits, seg = item / Cfg::kUnits;
    const int half = unit % Cfg::kHalves, sg = unit / Cfg::kHalves;
    const int cbase = sg * Cfg::kSlices * 32;
    const int first_real = (Cfg::kHalves == 2 && half == 1) ? 0 : Cfg::kHalo;
    const int col_base = (Cfg::kHalves == 2 && half == 1) ? Cfg::kPx - 3 : -Cfg::kHalo;   // column of slot 0

    char* const ring = smem + wave * Cfg::kWaveLds;
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;

    // ---- compute roles: 16 lanes x float2 = one pixel's 32-channel slice ----
    const int g = lane >> 4, l16 = lane & 15;
    const int strip = g % Cfg::kStrips, sl_i = g / Cfg::kStrips;
    const int ch = cbase + sl_i * 32 + 2 * l16;
    DwColState<W, BF> st;
#pragma unroll
    for (int t = 0; t < 49; ++t) st.wt[t] = *reinterpret_cast<const dwc_f32x2*>(wt + t * C + ch);
    st.bias = *reinterpret_cast<const dwc_f32x2*>(bias + ch);
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int p = 0; p < 7; ++p) st.acc[a][p] = st.bias;
    const char* const rd = ring + sl_i * Cfg::kSlots * Cfg::kSlotB + strip * 7 * Cfg::kSlotB + l16 * (BF ? 4 : 8);
    const unsigned yoff = (unsigned)(((half * Cfg::kPx + strip * 7) * C + ch) * kEsz);

    // ---- DMA roles: a 1-KB piece = kSPP pixel slots, kLanesPerSlot lanes x 16 B each ----
    unsigned voff[Cfg::kPieces];
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) {
        const int sl = p / Cfg::kPiecesPerSlice, k = p % Cfg::kPiecesPerSlice;
        int col = col_base + first_real + k * Cfg::kSPP + lane / Cfg::kLanesPerSlot;
        col = col < 0 ? 0 : (col >= W ? W - 1 : col);                 // masked lanes: any valid address
        voff[p] = (unsigned)((col * C + cbase + sl * 32) * kEsz + (lane % Cfg::kLanesPerSlot) * 16);
    }

    // ---- the segment: output rows [vb, ve) of the stacked image, input rows [vb - 3, ve + 3) ----
    const int Hp = H + 3;
    const long long Vt = (long long)B * Hp - 3;
    const int vb = (int)(Vt * seg / n_seg), ve = (int)(Vt * (seg + 1) / n_seg);
    const int steps = ve - vb + 6;
    if (ve <= vb) return;
    const char* const x0 = reinterpret_cast<const char*>(x_);
    // prefetch cursor: stacked row vb - 3 (may be -3 .. -1: rows above the first clip)
    int r_pf;                                                       // row inside its clip (>= H: one of the 3 zero rows)
    const char* pf_ptr;                                             // next in-image row at or after the cursor
    {
        const int v0 = vb - 3 + Hp;                                 // >= 0
        const int n1 = v0 / Hp;
        r_pf = v0 - n1 * Hp;
        const long long rows_before = (long long)(n1 - 1) * H + (r_pf < H ? r_pf : H);
        pf_ptr = x0 + rows_before * Cfg::kGRowB;
    }
    char* out_ptr;                                                  // next in-image output row at or after vb
    {
        const int n = vb / Hp, r = vb - n * Hp;
        out_ptr = reinterpret_cast<char*>(y_) + ((long long)n * H + (r < H ? r : H)) * Cfg::kGRowB;
    }
    char* const sink = reinterpret_cast<char*>(sink_);
    unsigned hist = 0;                                              // bit i: was the row prefetched i + 1 rows ago in-image?
    int u_pf = 0;                                                   // rows prefetched so far

    // zero the ring (the out-of-image column slots stay zero for good), then the first kD rows
    for (int i = lane; i < Cfg::kWaveLds / 16; i += 64) reinterpret_cast<dwc_f32x4*>(ring)[i] = dwc_f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define ACX_DWC_PREFETCH(slot_)                                                                                 \
    {                                                                                                           \
        const bool real_ = r_pf < H && u_pf < steps;                                                            \
        hist = (hist << 1) | (real_ ? 1u : 0u);                                                                 \
        const char* src_ = real_ ? pf_ptr : x0;                                                                 \
        pf_ptr += real_ ? Cfg::kGRowB : 0;                                                                      \
        r_pf = r_pf + 1 == Hp ? 0 : r_pf + 1;                                                                   \
        ++u_pf;                                                                                                 \
        dwc_issue_pieces<W, BF, 0>(src_, voff, ring_lds + (unsigned)((slot_) * Cfg::kRowB), first_real);        \
    }
#pragma unroll
    for (int i = 0; i < Cfg::kD; ++i) ACX_DWC_PREFETCH(i)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // once: the steady state below counts whole steps

    int j = 0;                                           // step = input row vb - 3 + j, output row vb - 6 + j
#define ACX_DWC_STEP(I_)                                                                                        \
    {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ACX_DWC_ABLATE >= 2 ? 0 : Cfg::kWait) : "memory");           \
        ACX_DWC_PREFETCH(((I_) + Cfg::kD) % 7)                                                                  \
        const bool in_real_ = (hist >> Cfg::kD) & 1u;                                                           \
        dwc_row<W, BF, I_>(st, rd, in_real_);                                                                   \
        const bool out_ok_ = ((hist >> (Cfg::kD + 3)) & 1u) && j >= 6;                                          \
        char* dst_ = (out_ok_ ? out_ptr : sink) + yoff;                                                         \
        out_ptr += out_ok_ ? Cfg::kGRowB : 0;                                                                   \
        _Pragma("unroll") for (int p = 0; p < (ACX_DWC_ABLATE == 3 ? 0 : 7); ++p) {                             \
            if constexpr (BF) *reinterpret_cast<unsigned*>(dst_ + p * C * kEsz) = acx_pack_bf16x2(st.acc[I_][p].x, st.acc[I_][p].y); \
            else *reinterpret_cast<dwc_f32x2*>(dst_ + p * C * kEsz) = st.acc[I_][p];                            \
        }                                                                                                       \
        if (++j == steps) break;                                                                                \
    }
    for (;;) {
        ACX_DWC_STEP(0) ACX_DWC_STEP(1) ACX_DWC_STEP(2) ACX_DWC_STEP(3) ACX_DWC_STEP(4) ACX_DWC_STEP(5) ACX_DWC_STEP(6)
    }
#undef ACX_DWC_STEP
#undef ACX_DWC_PREFETCH
}

// target_waves: how many waves the launch should consist of (one per SIMD of the CUs it may use)
template <int W, bool BF>
static int launch_dw_col_cfg(const void* x, void* y, const float* wt, const float* bias, void* sink, int B, int H,
                             int target_waves, hipStream_t s) {
    using Cfg = DwColCfg<W, BF>;
    const long long Vt = (long long)B * (H + 3) - 3;
    if (Vt > 0x7fffffffll) ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: batch too tall for one launch (%d clips of %d rows)", B, H);
    long long n_seg = target_waves / Cfg::kUnits;
    if (n_seg > Vt / kDwColMinRows) n_seg = Vt / kDwColMinRows;
    if (n_seg < 1) n_seg = 1;
    const int n_items = (int)(n_seg * Cfg::kUnits);
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &dwconv7_col_kernel<W, BF>, Cfg::kLdsBytes));
    dwconv7_col_kernel<W, BF><<<dim3((unsigned)((n_items + 3) / 4)), dim3(256), Cfg::kLdsBytes, s>>>(
        x, y, wt, bias, sink, B, H, (int)n_seg, n_items);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

int launch_dwconv_col(const void* x, void* y, const float* wt, const float* bias, void* sink, int B, int H, int W,
                      bool act_bf16, int target_waves, hipStream_t s) {
    switch (W) {
        case 56: return act_bf16 ? launch_dw_col_cfg<56, true>(x, y, wt, bias, sink, B, H, target_waves, s)
                                 : launch_dw_col_cfg<56, false>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 28: return act_bf16 ? launch_dw_col_cfg<28, true>(x, y, wt, bias, sink, B, H, target_waves, s)
                                 : launch_dw_col_cfg<28, false>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 14: return act_bf16 ? launch_dw_col_cfg<14, true>(x, y, wt, bias, sink, B, H, target_waves, s)
                                 : launch_dw_col_cfg<14, false>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 7: if (!act_bf16) return launch_dw_col_cfg<7, false>(x, y, wt, bias, sink, B, H, target_waves, s);
                ACX_FAIL(ACX_ERR_STATE, "dwconv7: stage 3 keeps fp32 activations");
        default: ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: unsupported width %d (expected 56/28/14/7)", W);
    }
}

}  // namespace acx
