// K3 -- depthwise 7x7 convolution, zero pad 3, + bias, channels-last (Block.dwconv,
// convnext.py:58-60 called at :76), and the per-pixel LayerNorm statistics that Block.norm
// (convnext.py:61,78; F.layer_norm :532-535) needs.
//
// HBM-bound: algorithmic bytes = read x + write y = 2*C*H*W*4 B per clip per block
// (10.84 / 5.42 / 2.71 / 1.33 MB in stages 0-3).  12.25 FLOP/B: the fp32 FMAs (0.945 GFLOP per clip) take about
// as long as the bytes, so everything else has to stay out of their way.  Lanes run along C (16 lanes x float2 =
// one 128-B line per pixel), each thread keeps 2 adjacent output rows x WT=7 adjacent pixels x 2 channels in
// registers; an input row (13 LDS reads) feeds both output rows, a kernel row's 7 weights are read once for the
// pair.  The input rows and the 49x32 weight slice are staged through LDS; the batch is streamed as ONE tall
// image (see the kernel), every input element is fetched from HBM once per column strip.
#include <type_traits>

#include "acx_internal.h"

namespace acx {

constexpr int kDwSlice = 32;      // channels per workgroup (staged as 8 x float4, computed as 16 lanes x float2)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ACX_STAMP(var)

// Streaming form: a workgroup owns (32-channel slice, column strip of TW pixels, segment of row tiles of the
// stacked batch) and walks DOWN the images.  An LDS ring of TH+6 input rows is kept; each step computes TH output rows from
// the ring while the next TH input rows are already in flight to registers (issued before the FMAs, written
// into the ring slots of the TH oldest rows after them), so every input row is fetched once per strip
// (the first tile-only version re-read its 6 halo rows per tile: FETCH_SIZE 1.93x the algorithmic bytes,
// profiles/r01_c_pmc_per_kernel.csv) and HBM latency hides under the arithmetic.
// BF: x and y are bf16 in HBM (ACX_PREC_BF16_ACT, stages 0-2): a 16-byte load carries 8 channels (a pixel's 32-channel
// slice is 64 bytes), widened to fp32 on its way into the LDS ring; outputs are rounded to bf16 (nearest even) when they
// leave.  Ring, weights, bias and all arithmetic stay fp32.
template <int TW, int TH, bool BF = false>
struct DwCfg {
    static constexpr int WT = 7;
    static constexpr int kStrips = TW / WT;
    static constexpr int kThreads = kStrips * 8 * TH;             // = strips x 16 lanes x TH/2 row pairs
    static constexpr int kCols = TW + 6;
    static constexpr int kRing = TH + 6;
    static constexpr int kRowF4 = kCols * 8;                      // float4 per ring row
    static constexpr int kEPV = BF ? 8 : 4;                       // tensor elements per 16-byte load
    static constexpr int kQ = kDwSlice / kEPV;                    // 16-byte loads per pixel slice
    static constexpr int kStepF4 = TH * kCols * kQ;               // 16-byte loads fetched per step
    static constexpr int kStage = (kStepF4 + kThreads - 1) / kThreads;   // staging loads per thread
    static constexpr size_t kLdsBytes = (size_t)(kRing * kRowF4 + 49 * 8 + 256) * 16;   // ring + weights + sinks
};

template <int TW, int TH, bool BF>
__global__ __launch_bounds__(256, 2) void dwconv7_kernel(const void* __restrict__ x_, void* __restrict__ y_,
                                                         const float* __restrict__ wt /*[49][C]*/,
                                                         const float* __restrict__ bias, int B, int H, int W,
                                                         int C, int tiles_w, int tiles_h, int n_seg,
                                                         unsigned magic /* floor(2^32 / (H + 3)) + 1 */) {
    using Cfg = DwCfg<TW, TH, BF>;
    using T = typename std::conditional<BF, __bf16, float>::type;
    const T* __restrict__ x = reinterpret_cast<const T*>(x_);
    T* __restrict__ y = reinterpret_cast<T*>(y_);
    static_assert(Cfg::kThreads == 256, "thread mapping assumes 256 threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* ring = reinterpret_cast<f32x4*>(smem);                  // [kRing][kCols][8]
    f32x4* wl = ring + Cfg::kRing * Cfg::kRowF4;                   // [49][8]
    f32x4* dummy = wl + 49 * 8;                                    // [256] per-thread sink for out-of-image columns

    int bid = blockIdx.x;
    const int slice = bid % (C / kDwSlice); bid /= (C / kDwSlice);
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int seg = bid;
    const int c0 = slice * kDwSlice;
    const int w0 = tw * TW;
    const int tid = threadIdx.x;
    const int t_begin = (int)((long long)tiles_h * seg / n_seg);     // balanced segments of row tiles
    const int t_end = (int)((long long)tiles_h * (seg + 1) / n_seg);
    if (t_begin >= t_end) return;

    // Prologue: ONE exposed memory latency.  The weight slice and both rounds of the first TH + 6 ring rows are
    // requested back to back, the ring is zeroed underneath them, and only then does anything wait (three
    // dependent round trips -- weights, rows, rows -- cost ~10 us per workgroup; stages 2-3 are little else).
    f32x4 wreg[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int i = tid + k * Cfg::kThreads;
        if (i >= 49 * 8) i = 49 * 8 - 1;
        wreg[k] = *reinterpret_cast<const f32x4*>(wt + (i >> 3) * C + c0 + 4 * (i & 7));
    }
    const T* xb = x + c0;
    // The B images of the batch form ONE tall virtual image: H rows of clip 0, 3 rows of zeros, H rows of clip 1,
    // ... (3 zero rows are all the 7x7 window ever sees between two clips, and consecutive clips are
    // consecutive in memory), so the ring streams straight through the batch and a workgroup lives for many
    // tiles whatever the image height -- per-image workgroups spent most of their life in ring prologue and
    // store drain once H fell to 63 and 31 rows.  Virtual row v = n * (H + 3) + r is image n, row r (r >= H: gap).
    const int Hp = H + 3;
#define ACX_DW_VROW(v_, n_, r_, ok_)   /* v_ may be negative (rows above the first clip) */                  \
    int n_, r_; bool ok_;                                                                                 \
    {                                                                                                     \
        const int vc_ = (v_) < 0 ? 0 : (v_);                                                              \
        n_ = (int)__umulhi((unsigned)vc_, magic);                                                         \
        r_ = vc_ - n_ * Hp;                                                                               \
        ok_ = (v_) >= 0 && r_ < H && n_ < B;                                                              \
    }
    const long long row_elems = (long long)W * C;

    // Per-thread staging plan for a step of TH image rows (float4 #k of this thread):
    //   st_ptr[k]  source of (row st_row[k], column, quad) for image row 0   (columns clamped into the image)
    //   st_lds[k]  float4 index inside a ring row, or -1 - (dummy index) when the column is outside the image
    // Loads are UNCONDITIONAL on valid addresses (a per-element "load or zero" makes hipcc branch around every
    // load and wait for all of them on the spot); rows outside the image only occur in the first/last step of a
    // column and take the slow (clamp + zero) path, selected by a wave-uniform test.
    const T* st_ptr[Cfg::kStage];
    int st_row[Cfg::kStage], st_lds[Cfg::kStage];
#pragma unroll
    for (int k = 0; k < Cfg::kStage; ++k) {
        int i = tid + k * Cfg::kThreads;
        const bool in_step = i < Cfg::kStepF4;
        if (!in_step) i = Cfg::kStepF4 - 1;
        const int qq = i % Cfg::kQ;
        const int col = (i / Cfg::kQ) % Cfg::kCols;
        st_row[k] = (i / Cfg::kQ) / Cfg::kCols;
        const int gw = w0 - 3 + col;
        const int gwc = gw < 0 ? 0 : (gw >= W ? W - 1 : gw);
        st_ptr[k] = xb + st_row[k] * row_elems + gwc * C + Cfg::kEPV * qq;
        st_lds[k] = (in_step && gw >= 0 && gw < W) ? (col * 8 + qq * (8 / Cfg::kQ)) : -1;     // BF: two float4 from here
    }
    const int g_origin = t_begin * TH - 3;             // image row kept in ring row 0 of this segment
    f32x4 rg[Cfg::kStage];
#define ACX_DW_LOAD_FAST(real_row)   /* TH rows of one clip, starting at row real_row of the (B*H)-row tensor */ \
    {                                                                                                     \
        const long long roff = (long long)(real_row) * row_elems;                                         \
        _Pragma("unroll") for (int k = 0; k < Cfg::kStage; ++k)                                           \
            rg[k] = *reinterpret_cast<const f32x4*>(st_ptr[k] + roff);                                    \
    }
#define ACX_DW_LOAD_EDGE_TO(rg_, first_row)   /* some rows outside [0,H): clamp the row, zero at store time */ \
    _Pragma("unroll") for (int k = 0; k < Cfg::kStage; ++k) {                                             \
        ACX_DW_VROW((first_row) + st_row[k], n_, r_, ok_)                                                 \
        const int gh = (n_ < B ? n_ : B - 1) * H + (r_ < H ? r_ : H - 1);     /* clamped: always a valid row */ \
        (void)ok_;                                                                                        \
        rg_[k] = *reinterpret_cast<const f32x4*>(st_ptr[k] + (long long)(gh - st_row[k]) * row_elems);    \
    }
#define ACX_DW_LOAD_EDGE(first_row) ACX_DW_LOAD_EDGE_TO(rg, first_row)
#define ACX_DW_STORE_FROM(rg_, first_row, max_rows, edge) /* registers -> ring rows of image rows first_row .. */ \
    {                                                                                                     \
        const int slot0 = ((first_row) - g_origin) % Cfg::kRing;           /* wave-uniform */             \
        _Pragma("unroll") for (int k = 0; k < Cfg::kStage; ++k) {                                         \
            int slot = slot0 + st_row[k];                                                                 \
            if (slot >= Cfg::kRing) slot -= Cfg::kRing;                                                   \
            f32x4 v = rg_[k];                                                                             \
            bool keep = st_lds[k] >= 0 && st_row[k] < (max_rows);                                         \
            if (edge) {                                                                                   \
                ACX_DW_VROW((first_row) + st_row[k], n_, r_, ok_)                                         \
                if (!ok_) v = f32x4{0.f, 0.f, 0.f, 0.f};                                                  \
            }                                                                                             \
            f32x4* dst = keep ? ring + slot * Cfg::kRowF4 + st_lds[k] : dummy + tid;                      \
            if (BF) {                               /* 8 bf16 -> two float4 (channels 8 qq .. 8 qq + 7) */   \
                const uint4 u_ = __builtin_bit_cast(uint4, v);                                            \
                dst[0] = f32x4{acx_bf16_lo(u_.x), acx_bf16_hi(u_.x), acx_bf16_lo(u_.y), acx_bf16_hi(u_.y)};   \
                (keep ? dst + 1 : dst)[0] = f32x4{acx_bf16_lo(u_.z), acx_bf16_hi(u_.z), acx_bf16_lo(u_.w), acx_bf16_hi(u_.w)}; \
            } else {                                                                                      \
                *dst = v;                                                                                 \
            }                                                                                             \
        }                                                                                                 \
    }
#define ACX_DW_STORE(first_row, max_rows, edge) ACX_DW_STORE_FROM(rg, first_row, max_rows, edge)
    // image rows [g_origin, g_origin + TH + 6) in two rounds (of the second only 6 rows are kept)
    {
        f32x4 rg2[Cfg::kStage];
        ACX_DW_LOAD_EDGE(g_origin)
        ACX_DW_LOAD_EDGE_TO(rg2, g_origin + TH)
        __builtin_amdgcn_sched_barrier(0);
        // columns outside the image are never written again: zero the whole ring once
        for (int i = tid; i < Cfg::kRing * Cfg::kRowF4; i += Cfg::kThreads) ring[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                               // ring zeroed before it is filled
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (tid + k * Cfg::kThreads < 49 * 8) wl[tid + k * Cfg::kThreads] = wreg[k];
        ACX_DW_STORE(g_origin, TH, true)
        ACX_DW_STORE_FROM(rg2, g_origin + TH, 6, true)
        __syncthreads();
    }

    // Compute mapping: 16 lanes x float2 cover a pixel's 32 channels; a thread owns TWO adjacent output rows x
    // WT = 7 adjacent pixels.  An input row read from LDS (13 float2) then feeds both output rows (kernel rows
    // ky and ky-1), and a kernel row's 7 weights are read once for the pair: 8*13 + 49 = 153 8-byte reads per
    // 28 outputs instead of 2 x (7*13 + 49) 16-byte reads per 2 x 28 with one row per thread -- the LDS pipe
    // (128 B/clk per CU), not the VALU, was what bounded the FMA phase (4.5 k LDS cycles against 2.7 k of
    // v_pk_fma_f32 per workgroup step).
    const int l16 = tid & 15;
    const int strip = (tid >> 4) % Cfg::kStrips;
    const int rp = (tid >> 4) / Cfg::kStrips;          // row pair: output rows h0 + 2rp, h0 + 2rp + 1
    const f32x2* const ring2 = reinterpret_cast<const f32x2*>(ring);
    const f32x2* const wl2 = reinterpret_cast<const f32x2*>(wl);       // [49][16]
    constexpr int kRowF2 = Cfg::kRowF4 * 2;
    const f32x2 bv = *reinterpret_cast<const f32x2*>(bias + c0 + 2 * l16);
    const int rd_off = (strip * Cfg::WT) * 16 + l16;   // float2 offset of this thread's first input column
    T* const yb = y + (long long)(w0 + strip * Cfg::WT) * C + c0 + 2 * l16;

    f32x2 so0[Cfg::WT], so1[Cfg::WT];
#pragma unroll
    for (int i = 0; i < Cfg::WT; ++i) so0[i] = so1[i] = bv;
    T *yp0 = nullptr, *yp1 = nullptr;                  // where so0[] / so1[] belong (null: nothing pending)
#define ACX_DW_FLUSH1(yp_, so_) if (yp_ != nullptr) { _Pragma("unroll") for (int i = 0; i < Cfg::WT; ++i) {         \
        if (BF) *reinterpret_cast<unsigned*>(yp_ + (long long)i * C) = acx_pack_bf16x2(so_[i].x, so_[i].y);           \
        else *reinterpret_cast<f32x2*>(yp_ + (long long)i * C) = so_[i]; } }
#define ACX_DW_FLUSH ACX_DW_FLUSH1(yp0, so0) ACX_DW_FLUSH1(yp1, so1)
    for (int t = t_begin; t < t_end; ++t) {
        const int h0 = t * TH;
        const bool more = t + 1 < t_end;
        ACX_STAMP(ts0)
        const int next_first = h0 + TH + 3;            // image rows of the next step (ring rows of the TH oldest)
        // fast path: the TH rows of the next step belong to one clip (next_first >= 0 always)
        const int nf_n = (int)__umulhi((unsigned)next_first, magic);
        const int nf_r = next_first - nf_n * Hp;
        const bool edge = nf_r + TH > H || nf_n >= B;
        if (more) {
            if (edge) { ACX_DW_LOAD_EDGE(next_first) } else { ACX_DW_LOAD_FAST(nf_n * H + nf_r) }
        }
        __builtin_amdgcn_sched_barrier(0);
        // Outputs leave one tile LATE, from their own register set, right BEHIND the next prefetch: hipcc puts a
        // vmcnt(0) in front of the prefetch address set-up (it cannot prove the staging registers idle across
        // the back edge), so nothing may be in flight there; the one real wait of the loop (before the ring
        // refill) then covers loads and stores that both had a whole FMA phase to complete.
        ACX_DW_FLUSH
        __builtin_amdgcn_sched_barrier(0);

        f32x2 a0[Cfg::WT], a1[Cfg::WT];
#pragma unroll
        for (int i = 0; i < Cfg::WT; ++i) a0[i] = a1[i] = bv;
        int lw = l16;                     // opaque per tile: keeps hipcc from hoisting all 49 weight float2
        asm volatile("" : "+v"(lw));      // (98 VGPRs) out of the tile loop -- they are re-read from LDS instead
        const int base = (h0 - 3 - g_origin + 2 * rp) % Cfg::kRing;    // ring slot of input row h0 - 3 + 2rp
        // The 8 input rows are software-pipelined by hand in HALF rows: the LDS reads of the next unit
        // (7 or 6 input float2, and once per row the 7 weights of the next kernel row) are issued into
        // a second register set BEFORE the packed FMAs of the current unit.  Input row i meets kernel row i for
        // the upper output row and kernel row i-1 for the lower one: three weight sets rotate.
        f32x2 iA[7], iB[6], wA[7], wB[7], wC[7];
#define ACX_DW_ROWP(i_, p_)                                                                               \
        const f32x2* p_;                                                                                  \
        {                                                                                                 \
            int slot = base + (i_);                                                                       \
            if (slot >= Cfg::kRing) slot -= Cfg::kRing;                                                   \
            p_ = ring2 + slot * kRowF2 + rd_off;                                                          \
        }
#define ACX_DW_READ_W(w_, ky_) _Pragma("unroll") for (int kx = 0; kx < 7; ++kx) w_[kx] = wl2[((ky_) * 7 + kx) * 16 + lw];
#define ACX_DW_READ_I0(i_) { ACX_DW_ROWP(i_, p0_) _Pragma("unroll") for (int j = 0; j < 7; ++j) iA[j] = p0_[j * 16]; }
#define ACX_DW_READ_I1(i_) { ACX_DW_ROWP(i_, p1_) _Pragma("unroll") for (int j = 0; j < 6; ++j) iB[j] = p1_[(7 + j) * 16]; }
        // The packed FMAs are inline asm in a FIXED order.  Left to hipcc they are regrouped into per-accumulator
        // chains (v_pk_fma_f32 a, x0, w0, a; s_nop; v_pk_fma_f32 a, x1, w1, a; ...: 254 hazard nops per tile and a
        // dependent instruction every slot -- a wave could then fill at most half of its SIMD's issue slots and
        // the FMA phase ran at 0.54 of the measured v_pk_fma_f32 rate); here the same accumulator returns after
        // >= 2 (single output row) or >= 4 (both rows) other FMAs.  The LDS reads of the NEXT unit are issued in
        // adjacent-column pairs (one ds_read2_b64 each) at even distances inside the FMA stream, fenced by
        // sched_barrier so that they stay where they are put.
#define ACX_DW_PKFMA(acc_, x_, w_) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc_) : "v"(x_), "v"(w_));
#define ACX_DW_UNIT(i_, wc_, wp_, in_, jlo_, jhi_, nslots_, slot_fn_)                                      \
        {                                                                                                 \
            constexpr int both_ = ((i_) >= 1 && (i_) <= 6) ? 2 : 1;                                       \
            constexpr int total_ = ((jlo_) == 0 ? 28 : 21) * both_;                                       \
            constexpr int gap_ = total_ / ((nslots_) + 1) > 0 ? total_ / ((nslots_) + 1) : 1;             \
            int n_ = 0, sl_ = 0;                                                                          \
            _Pragma("unroll") for (int j = (jlo_); j < (jhi_); ++j)                                       \
            _Pragma("unroll") for (int kx = (j > 6 ? j - 6 : 0); kx <= (j > 6 ? 6 : j); ++kx) {           \
                if ((i_) <= 6) { ACX_DW_PKFMA(a0[j - kx], in_[j - (jlo_)], wc_[kx]) ++n_; }               \
                if ((i_) >= 1) { ACX_DW_PKFMA(a1[j - kx], in_[j - (jlo_)], wp_[kx]) ++n_; }               \
                if (sl_ < (nslots_) && n_ >= (sl_ + 1) * gap_) {                                          \
                    __builtin_amdgcn_sched_barrier(0);                                                    \
                    slot_fn_(sl_);                                                                        \
                    ++sl_;                                                                                \
                    __builtin_amdgcn_sched_barrier(0);                                                    \
                }                                                                                         \
            }                                                                                             \
            _Pragma("unroll") for (; sl_ < (nslots_); ++sl_) slot_fn_(sl_);                               \
        }
        // one asm statement per register set: ONE lgkmcnt wait, placed in front of the next batch of reads
#define ACX_DW_TOUCH6(a_) asm volatile("" :: "v"(a_[0]), "v"(a_[1]), "v"(a_[2]), "v"(a_[3]), "v"(a_[4]), "v"(a_[5]));
#define ACX_DW_TOUCH7(a_) asm volatile("" :: "v"(a_[0]), "v"(a_[1]), "v"(a_[2]), "v"(a_[3]), "v"(a_[4]), "v"(a_[5]), "v"(a_[6]));
        // input row i_: current kernel row in wc_ (upper output row), previous one in wp_ (lower output row),
        // the next one is prefetched into wn_
#define ACX_DW_IROW(i_, wc_, wp_, wn_)                                                                    \
        {                                                                                                 \
            ACX_DW_ROWP(i_, pB_)                                                                          \
            auto rdB_ = [&](int k) __attribute__((always_inline)) { iB[2 * k] = pB_[(7 + 2 * k) * 16]; iB[2 * k + 1] = pB_[(8 + 2 * k) * 16]; }; \
            ACX_DW_UNIT(i_, wc_, wp_, iA, 0, 7, 3, rdB_)                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            ACX_DW_TOUCH6(iB)                                                                             \
            ACX_DW_ROWP(((i_) + 1 < 8 ? (i_) + 1 : 0), pA_)                                               \
            const f32x2* const pW_ = wl2 + (((i_) + 1 < 7 ? (i_) + 1 : 0) * 7) * 16 + lw;                 \
            auto rdA_ = [&](int k) __attribute__((always_inline)) {              /* slots 0..3: next input row, 4..7: next kernel row */  \
                if (k < 3) { iA[2 * k] = pA_[(2 * k) * 16]; iA[2 * k + 1] = pA_[(2 * k + 1) * 16]; }       \
                else if (k == 3) { iA[6] = pA_[6 * 16]; }                                                 \
                else if (k < 7) { wn_[2 * (k - 4)] = pW_[(2 * (k - 4)) * 16]; wn_[2 * (k - 4) + 1] = pW_[(2 * (k - 4) + 1) * 16]; } \
                else { wn_[6] = pW_[6 * 16]; }                                                            \
            };                                                                                            \
            ACX_DW_UNIT(i_, wc_, wp_, iB, 7, 13, ((i_) + 1 < 7 ? 8 : ((i_) + 1 < 8 ? 4 : 0)), rdA_)       \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if ((i_) + 1 < 7) { ACX_DW_TOUCH7(wn_) }                                                      \
            if ((i_) + 1 < 8) { ACX_DW_TOUCH7(iA) }                                                       \
        }
        ACX_DW_READ_W(wA, 0)
        ACX_DW_READ_I0(0)
        ACX_DW_IROW(0, wA, wC, wB) ACX_DW_IROW(1, wB, wA, wC) ACX_DW_IROW(2, wC, wB, wA) ACX_DW_IROW(3, wA, wC, wB)
        ACX_DW_IROW(4, wB, wA, wC) ACX_DW_IROW(5, wC, wB, wA) ACX_DW_IROW(6, wA, wC, wB) ACX_DW_IROW(7, wB, wA, wC)
#undef ACX_DW_ROWP
#undef ACX_DW_READ_W
#undef ACX_DW_READ_I0
#undef ACX_DW_READ_I1
#undef ACX_DW_PKFMA
#undef ACX_DW_UNIT
#undef ACX_DW_TOUCH6
#undef ACX_DW_TOUCH7
#undef ACX_DW_IROW
        ACX_STAMP(ts1)
        __builtin_amdgcn_sched_barrier(0);
        ACX_STAMP(ts2)
        if (more) {
            __syncthreads();                                       // everyone is done reading the ring
            ACX_STAMP(ts3)
            asm volatile("" : "+v"(rg[0]));                        // keep the vmcnt wait down here
            if (edge) { ACX_DW_STORE(next_first, TH, true) } else { ACX_DW_STORE(next_first, TH, false) }
            ACX_STAMP(ts4)
            __syncthreads();
            ACX_STAMP(ts5)
        }
#pragma unroll
        for (int i = 0; i < Cfg::WT; ++i) { so0[i] = a0[i]; so1[i] = a1[i]; }
        {
            ACX_DW_VROW(h0 + 2 * rp, n0_, r0_, ok0_)
            ACX_DW_VROW(h0 + 2 * rp + 1, n1_, r1_, ok1_)
            yp0 = ok0_ ? yb + (long long)(n0_ * H + r0_) * row_elems : nullptr;
            yp1 = ok1_ ? yb + (long long)(n1_ * H + r1_) * row_elems : nullptr;
        }
    }
    ACX_DW_FLUSH
#undef ACX_DW_FLUSH
#undef ACX_DW_FLUSH1
#undef ACX_DW_LOAD_FAST
#undef ACX_DW_VROW
#undef ACX_DW_LOAD_EDGE
#undef ACX_DW_LOAD_EDGE_TO
#undef ACX_DW_STORE
#undef ACX_DW_STORE_FROM
}

// Per-row LayerNorm statistics over C channels (biased variance, eps inside the sqrt --
// convnext.py:537-540 / F.layer_norm): stats[row] = (mean, rstd).  G = C/12 lanes per row, each
// lane holds 3 float4; reductions are xor-shuffles inside the G-lane group (G = 8..64).
// NORMALIZE 1: write (x - mean) * rstd instead (input of the downsample convs, convnext.py:230-235).
// NORMALIZE 2: the same rounded to bf16, rows padded with zeros to a multiple of 64 channels (A operand of the
// bf16-precision GEMMs, gemm_bf16.hip); the statistics and the normalisation itself stay fp32.
// NORMALIZE 3: the same in S16 form (gemm_split.hip): scaled by 2^11, each value as fp16 hi + fp16 lo, blocks of
// 8 channels = [hi x8][lo x8] -- the same 4 bytes per element as fp32.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int G, int NORMALIZE>
__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                       long long rows, float eps) {
    constexpr int C = G * 12;
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x % G;
    const int sub = threadIdx.x / G;
    const long long stride = (long long)gridDim.x * RPB;
    const long long iters = (rows + stride - 1) / stride;
    for (long long it = 0; it < iters; ++it) {
        long long row = it * stride + (long long)blockIdx.x * RPB + sub;
        const bool valid = row < rows;
        if (!valid) row = rows - 1;
        const float4* p = reinterpret_cast<const float4*>(x + row * C);
        float4 v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = p[g + G * k];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = v[k].x - mean, b2 = v[k].y - mean, c2 = v[k].z - mean, e = v[k].w - mean;
            d += (a * a + b2 * b2) + (c2 * c2 + e * e);
        }
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) d += __shfl_xor(d, o);
        const float rstd = 1.0f / sqrtf(d * (1.0f / C) + eps);
        if (NORMALIZE == 3) {
            if (valid) {
                char* o = reinterpret_cast<char*>(stats) + row * (C * 4);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int f4 = g + G * k;            // float4 index in the row: block f4>>1, half f4&1
                    f16x4 hi, lo;
                    const float sc = rstd * kSplitLnScale;
                    const float t[4] = {(v[k].x - mean) * sc, (v[k].y - mean) * sc, (v[k].z - mean) * sc, (v[k].w - mean) * sc};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (_Float16)t[e];
                        lo[e] = (_Float16)(t[e] - (float)hi[e]);
                    }
                    char* blk = o + (f4 >> 1) * 32 + (f4 & 1) * 8;
                    *reinterpret_cast<f16x4*>(blk) = hi;
                    *reinterpret_cast<f16x4*>(blk + 16) = lo;
                }
            }
        } else if (NORMALIZE == 2) {
            if (valid) {
                constexpr int Cp = (C + 63) / 64 * 64;
                __bf16* o = reinterpret_cast<__bf16*>(stats) + row * Cp;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    bf16x4 q;
                    q[0] = (__bf16)((v[k].x - mean) * rstd); q[1] = (__bf16)((v[k].y - mean) * rstd);
                    q[2] = (__bf16)((v[k].z - mean) * rstd); q[3] = (__bf16)((v[k].w - mean) * rstd);
                    *reinterpret_cast<bf16x4*>(o + 4 * (g + G * k)) = q;
                }
                if (Cp > C && 4 * g < Cp - C) {
                    bf16x4 z; z[0] = z[1] = z[2] = z[3] = (__bf16)0.f;
                    *reinterpret_cast<bf16x4*>(o + C + 4 * g) = z;
                }
            }
        } else if (NORMALIZE == 1) {      // `stats` is the (rows, C) output: (x - mean) * rstd, affine folded downstream
            if (valid) {
                float4* o = reinterpret_cast<float4*>(stats + row * C);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    o[g + G * k] = make_float4((v[k].x - mean) * rstd, (v[k].y - mean) * rstd,
                                               (v[k].z - mean) * rstd, (v[k].w - mean) * rstd);
            }
        } else if (valid && g == 0) {
            *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(mean, rstd);
        }
    }
}

template <int NORMALIZE>
static int launch_rows(acx_ctx* c, const float* x, float* out, int64_t M, int C, hipStream_t s) {
    const int G = C / 12;
    long long blocks = (M + (256 / G) - 1) / (256 / G);
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(c, ACX_K_ROWSTATS, s);
    dim3 grid((unsigned)blocks), blk(256);
    switch (C) {
        case 96: launch_kernel(&rowstats_kernel<8, NORMALIZE>, grid, blk, 0, s, x, out, M, 1e-6f); break;
        case 192: launch_kernel(&rowstats_kernel<16, NORMALIZE>, grid, blk, 0, s, x, out, M, 1e-6f); break;
        case 384: launch_kernel(&rowstats_kernel<32, NORMALIZE>, grid, blk, 0, s, x, out, M, 1e-6f); break;
        case 768: launch_kernel(&rowstats_kernel<64, NORMALIZE>, grid, blk, 0, s, x, out, M, 1e-6f); break;
        default: ACX_FAIL(ACX_ERR_SHAPE, "rowstats: unsupported channel count %d", C);
    }
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

int launch_rowstats(acx_ctx* c, const float* x, float* stats, int64_t M, int C, hipStream_t s) {
    return launch_rows<0>(c, x, stats, M, C, s);
}

int launch_layernorm_rows(acx_ctx* c, const float* x, float* out, int64_t M, int C, hipStream_t s) {
    return launch_rows<1>(c, x, out, M, C, s);
}

int launch_layernorm_rows_split(acx_ctx* c, const float* x, void* out, int64_t M, int C, hipStream_t s) {
    return launch_rows<3>(c, x, reinterpret_cast<float*>(out), M, C, s);
}

int launch_layernorm_rows_bf16(acx_ctx* c, const float* x, void* out, int64_t M, int C, hipStream_t s) {
    return launch_rows<2>(c, x, reinterpret_cast<float*>(out), M, C, s);
}

template <int TW, int TH, bool BF>
static int launch_dw_cfg(const BlockW& w, int C, const void* x, void* y, int B, int H, int W, hipStream_t s) {
    // exactness of v / (H + 3) by multiply-high needs (stacked rows) * (H + 3) < 2^32: longer batches in chunks
    const long long max_b = (0xffffffffll / (H + 3)) / (H + 3);
    if (B > max_b) {
        for (long long b0 = 0; b0 < B; b0 += max_b) {
            const int nb = (int)((B - b0) < max_b ? (B - b0) : max_b);
            const long long off = b0 * (long long)H * W * C * (BF ? 2 : 4);
            ACX_TRY((launch_dw_cfg<TW, TH, BF>(w, C, reinterpret_cast<const char*>(x) + off, reinterpret_cast<char*>(y) + off, nb, H, W, s)));
        }
        return ACX_OK;
    }
    using Cfg = DwCfg<TW, TH, BF>;
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &dwconv7_kernel<TW, TH, BF>, Cfg::kLdsBytes));
    const int Hv = B * (H + 3) - 3;                    // stacked rows (no gap after the last clip)
    const int tiles_w = W / TW, tiles_h = (Hv + TH - 1) / TH;
    const long long columns = (long long)tiles_w * (C / kDwSlice);
    // Workgroups = (column strips x channel slices) x row segments of the stacked image.  Two workgroups are
    // resident per CU (512 slots): ONE round of equal segments -- every further round pays the ring prologue
    // and the store drain again, and a fractional round idles the chip for a whole workgroup life.
    constexpr int kTargetWorkgroups = 512;      // one round: 2 workgroups per CU
    int n_seg = (int)(kTargetWorkgroups / columns);
    if (n_seg > tiles_h / 2) n_seg = tiles_h / 2;
    if (n_seg < 1) n_seg = 1;
    const long long blocks = columns * n_seg;
    launch_kernel(&dwconv7_kernel<TW, TH, BF>, dim3((unsigned)blocks), dim3(Cfg::kThreads), Cfg::kLdsBytes, s,
        x, y, w.dw, w.dwb, B, H, W, C, tiles_w, tiles_h, n_seg, (unsigned)(0x100000000ull / (unsigned)(H + 3)) + 1u);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// Stages 2 and 3 (images of 63 x 14 and 31 x 7 pixels per clip, tensors that sit in the Infinity Cache): the streaming
// kernel above gives a workgroup only 3-6 row tiles there, so its life is mostly ring prologue and store drain (52 / 37 us
// for 173 / 85 MB).  Here a workgroup takes ONE tile of ROWS output rows x the full width of one clip x one 32-channel slice:
// the tile with its halo (zero outside the image) is loaded in one go -- one memory round trip in, one out, nothing to
// stream; the 6 halo rows a tile re-reads come from the cache (W = 14: 22 rows for 16; W = 7 and H <= 32: none).
// Thread (lane16 = channel pair, g = 0..15 = (row pair, column strip)) computes 2 rows x 7 pixels x 2 channels as in the
// streaming kernel: an input row (13 reads) feeds both output rows, a kernel row's 7 weights are read once for the pair.
// W = 7: the two groups of a half-wave read rows two apart -- the row stride is padded by 64 B so that 2 strides = 128
// (mod 256): disjoint banks; W = 14: they read the two strips of one row, 896 B apart: disjoint as it is.
// BF: x and y are bf16 (ACX_PREC_BF16_ACT, stage 2): 16-byte loads carry 8 channels, widened on their way into the LDS.
template <int W, bool BF>
struct DwTileCfg {
    static constexpr int kStrips = W / 7;
    static constexpr int kRows = 32 / kStrips;                   // output rows per tile: 16 groups = row pairs x strips
    static constexpr int kCols = W + 6;
    static constexpr int kRowBytes = kCols * 128 + (kStrips == 1 ? 64 : 0);
    static constexpr int kEPV = BF ? 8 : 4;                      // tensor elements per 16-byte load
    static constexpr int kQ = kDwSlice / kEPV;                   // 16-byte loads per pixel slice
    static constexpr int kLoads = ((kRows + 6) * W * kQ + 255) / 256;     // per thread
    static constexpr size_t kLdsBytes = (size_t)(kRows + 6) * kRowBytes + 49 * 128;
};

template <int W, bool BF>
__global__ __launch_bounds__(256, 2) void dwconv7_tile_kernel(const void* __restrict__ x_, void* __restrict__ y_,
                                                              const float* __restrict__ wt /*[49][C]*/,
                                                              const float* __restrict__ bias, int H, int C, int tiles_h) {
    using Cfg = DwTileCfg<W, BF>;
    using T = typename std::conditional<BF, __bf16, float>::type;
    constexpr int kRowBytes = Cfg::kRowBytes, kIn = Cfg::kRows + 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* img = smem;                                                           // [kIn][Cfg::kCols][32] fp32, rows kRowBytes apart
    f32x4* wl = reinterpret_cast<f32x4*>(smem + kIn * kRowBytes);               // [49][8]
    const int slices = C / kDwSlice;
    int bid = blockIdx.x;
    const int slice = bid % slices; bid /= slices;
    const int th = bid % tiles_h;
    const long long b = bid / tiles_h;
    const int c0 = slice * kDwSlice;
    const int r0 = th * Cfg::kRows;                                             // first output row of the tile
    const int tid = threadIdx.x;
    const T* xb = reinterpret_cast<const T*>(x_) + b * (long long)H * W * C + c0;
    // every load of the workgroup is requested before anything waits: the tile's rows r0 - 3 .. r0 + kRows + 2 (rows outside
    // the image: clamped address, zeroed below) and the 49 x 32 weights
    f32x4 v[Cfg::kLoads], wreg[2];
#pragma unroll
    for (int k = 0; k < Cfg::kLoads; ++k) {
        int i = tid + 256 * k;
        if (i >= kIn * W * Cfg::kQ) i = kIn * W * Cfg::kQ - 1;
        const int q = i % Cfg::kQ, px = i / Cfg::kQ, lr = px / W, cidx = px - lr * W;
        int r = r0 - 3 + lr;
        r = r < 0 ? 0 : (r >= H ? H - 1 : r);
        v[k] = *reinterpret_cast<const f32x4*>(xb + (long long)(r * W + cidx) * C + Cfg::kEPV * q);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int i = tid + 256 * k;
        if (i >= 49 * 8) i = 49 * 8 - 1;
        wreg[k] = *reinterpret_cast<const f32x4*>(wt + (i >> 3) * C + c0 + 4 * (i & 7));
    }
    // the zero columns left and right of the image while the loads fly (rows outside the image are zeroed with their data)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < kIn * 6 * 8; i += 256) {
        const int q = i & 7, cb = (i >> 3) % 6, lr = (i >> 3) / 6;
        const int col = cb < 3 ? cb : W + cb;                                   // 0, 1, 2, W + 3, W + 4, W + 5
        *reinterpret_cast<f32x4*>(img + lr * kRowBytes + col * 128 + 16 * q) = zero;
    }
#pragma unroll
    for (int k = 0; k < Cfg::kLoads; ++k) {
        const int i = tid + 256 * k;
        if (i < kIn * W * Cfg::kQ) {
            const int q = i % Cfg::kQ, px = i / Cfg::kQ, lr = px / W, cidx = px - lr * W;
            const int r = r0 - 3 + lr;
            const bool inside = r >= 0 && r < H;
            char* dst = img + lr * kRowBytes + (cidx + 3) * 128;
            if (BF) {                                   // 8 bf16 -> two float4 (channels 8 q .. 8 q + 7)
                const uint4 u = __builtin_bit_cast(uint4, v[k]);
                f32x4 lo4 = {acx_bf16_lo(u.x), acx_bf16_hi(u.x), acx_bf16_lo(u.y), acx_bf16_hi(u.y)};
                f32x4 hi4 = {acx_bf16_lo(u.z), acx_bf16_hi(u.z), acx_bf16_lo(u.w), acx_bf16_hi(u.w)};
                *reinterpret_cast<f32x4*>(dst + 32 * q) = inside ? lo4 : zero;
                *reinterpret_cast<f32x4*>(dst + 32 * q + 16) = inside ? hi4 : zero;
            } else {
                *reinterpret_cast<f32x4*>(dst + 16 * q) = inside ? v[k] : zero;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 256 * k;
        if (i < 49 * 8) wl[i] = wreg[k];
    }
    __syncthreads();

    const int l16 = tid & 15, g = tid >> 4;
    const int rp = g / Cfg::kStrips, strip = g % Cfg::kStrips;                  // output rows r0 + 2 rp, + 1; pixels 7 strip .. + 6
    const f32x2 bv = *reinterpret_cast<const f32x2*>(bias + c0 + 2 * l16);
    f32x2 acc[2][7];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int p = 0; p < 7; ++p) acc[o][p] = bv;
    const char* in0 = img + (2 * rp) * kRowBytes + (7 * strip) * 128 + 8 * l16;  // tile row 2 rp = image row r0 + 2 rp - 3
    const f32x2* wrow = reinterpret_cast<const f32x2*>(wl) + l16;               // tap t at wrow[16 t]
    // input row 2 rp + ky of the tile meets kernel row ky of output row 2 rp (ky < 7) and kernel row ky - 1 of output row
    // 2 rp + 1 (ky >= 1).  The six middle rows run as a rolled loop: unrolled, hipcc hoists all 104 row reads and 98 weight
    // reads to the top and spills.
#define ACX_DWI_ROW(ky_, do0_, do1_)                                                                            \
    {                                                                                                           \
        f32x2 in[13];                                                                                           \
        _Pragma("unroll") for (int cidx = 0; cidx < 13; ++cidx)                                                 \
            in[cidx] = *reinterpret_cast<const f32x2*>(in0 + (ky_) * kRowBytes + cidx * 128);                   \
        if (do0_) {                                                                                             \
            _Pragma("unroll") for (int kx = 0; kx < 7; ++kx) {                                                  \
                const f32x2 wv = wrow[16 * (7 * (ky_) + kx)];                                                   \
                _Pragma("unroll") for (int p = 0; p < 7; ++p) acc[0][p] = in[p + kx] * wv + acc[0][p];          \
            }                                                                                                   \
        }                                                                                                       \
        if (do1_) {                                                                                             \
            _Pragma("unroll") for (int kx = 0; kx < 7; ++kx) {                                                  \
                const f32x2 wv = wrow[16 * (7 * ((ky_) - 1) + kx)];                                             \
                _Pragma("unroll") for (int p = 0; p < 7; ++p) acc[1][p] = in[p + kx] * wv + acc[1][p];          \
            }                                                                                                   \
        }                                                                                                       \
    }
    ACX_DWI_ROW(0, true, false)
#pragma nounroll
    for (int ky = 1; ky < 7; ++ky) ACX_DWI_ROW(ky, true, true)
    ACX_DWI_ROW(7, false, true)
#undef ACX_DWI_ROW
    T* yb = reinterpret_cast<T*>(y_) + b * (long long)H * W * C + c0 + 2 * l16;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int r = r0 + 2 * rp + o;
        if (r < H) {
#pragma unroll
            for (int p = 0; p < 7; ++p) {
                T* dst = yb + (long long)(r * W + 7 * strip + p) * C;
                if (BF) *reinterpret_cast<unsigned*>(dst) = acx_pack_bf16x2(acc[o][p].x, acc[o][p].y);
                else *reinterpret_cast<f32x2*>(dst) = acc[o][p];
            }
        }
    }
}

// ONE tile per short-lived workgroup.  Measured against long-lived workgroups walking runs of ~6 tiles with the next tile's
// rows prefetched into registers (one round of 512 workgroups): 47 vs 56 us per stage-2 launch at B = 64 -- with two
// workgroups per CU the hardware overlaps one's load phase with the other's arithmetic better than a software pipeline with
// two barriers per tile does.
template <int W, bool BF>
static int launch_dw_tile(const BlockW& w, int C, const void* x, void* y, int B, int H, hipStream_t s) {
    using Cfg = DwTileCfg<W, BF>;
    const int tiles_h = (H + Cfg::kRows - 1) / Cfg::kRows;
    const long long per_clip = (long long)tiles_h * (C / kDwSlice);
    const long long max_b = 0x3fffffffll / per_clip;           // grid limit: longer batches in chunks of clips
    if (B > max_b) {
        for (long long b0 = 0; b0 < B; b0 += max_b) {
            const int nb = (int)((B - b0) < max_b ? (B - b0) : max_b);
            const long long off = b0 * (long long)H * W * C * (BF ? 2 : 4);
            ACX_TRY((launch_dw_tile<W, BF>(w, C, reinterpret_cast<const char*>(x) + off, reinterpret_cast<char*>(y) + off, nb, H, s)));
        }
        return ACX_OK;
    }
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &dwconv7_tile_kernel<W, BF>, Cfg::kLdsBytes));
    launch_kernel(&dwconv7_tile_kernel<W, BF>, dim3((unsigned)(B * per_clip)), dim3(256), Cfg::kLdsBytes, s, x, y, w.dw, w.dwb, H, C, tiles_h);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// The column-streaming kernel (dwconv_col.hip) takes the launches whose waves get segments long enough to amortise their six
// halo rows and their prologue: one wave per SIMD of the CUs this launch may count on (all of them, or its share while
// acx_forward runs several sub-batches side by side).  Same bits either way (same accumulation order per output element);
// ACX_DW_STREAM = 0 | 1 forces a form (tests).
static bool use_col_kernel(const acx_ctx* c, int B, int H, int W, bool act_bf16, int* target_waves) {
    if (!c || !c->d_dw_sink) return false;
    const int force = tuning().dw_stream.load(std::memory_order_relaxed);
    if (force == 0) return false;
    int cus = 0;
    if (cu_count_of_current_device(&cus) != ACX_OK) return false;
    *target_waves = 4 * cus / inflight_ways();
    const long long Vt = (long long)B * (H + 3) - 3;
    if ((Vt + 16ll * (H + 3) + 64) * (H + 3) >= 0xffffffffll) return false;   // beyond the multiply-high division of the column kernel: the ring kernel chunks
    if (force == 1) return true;
    // rows of the stacked batch per wave segment (dwconv_col.hip: one wave per SIMD in the fp32 stages 0-1, two elsewhere)
    const bool two = act_bf16 || W <= 14;
    const long long segs = (long long)*target_waves * (two ? 2 : 1) / 6;
    return Vt / (segs > 0 ? segs : 1) >= (two ? (W == 7 ? 6 : 10) : 40);
}

int launch_dwconv(acx_ctx* c, const BlockW& w, int C, const void* x, void* y, float* stats, int B, int H,
                  int W, hipStream_t s, bool act_bf16) {
    if (act_bf16 && stats) ACX_FAIL(ACX_ERR_STATE, "dwconv7: row statistics are computed from fp32 activations only");
    {
        ProfScope ps(c, ACX_K_DWCONV, s);
        int rc, target_waves = 0;
        // bf16 activations (stages 0-2 of "bf16a"): the matrix-pipe kernel, at every launch size -- its arithmetic (bf16 weights,
        // its own summation order) must not depend on the batch
        if (act_bf16 && c && c->d_dw_sink && w.dw_ops && C == 96 * 56 / (W > 0 ? W : 1) && W != 7 && tuning().dw_mfma.load(std::memory_order_relaxed) != 0) {
            int cus = 0;
            ACX_TRY(cu_count_of_current_device(&cus));
            const int per_cu = tuning().dwm_waves.load(std::memory_order_relaxed);
            return launch_dwconv_mfma(x, y, w.dw_ops, w.dwb, c->d_dw_sink, B, H, W, (per_cu ? per_cu : 8) * cus / inflight_ways(), s);
        }
        if (C == 96 * 56 / (W > 0 ? W : 1) && !act_bf16 && use_col_kernel(c, B, H, W, act_bf16, &target_waves)) {
            rc = launch_dwconv_col(x, y, w.dw, w.dwb, c->d_dw_sink, B, H, W, act_bf16, target_waves, s);
            ACX_TRY(rc);
            if (stats) ACX_TRY(launch_rowstats(c, reinterpret_cast<const float*>(y), stats, (int64_t)B * H * W, C, s));
            return ACX_OK;
        }
        switch (W) {
            case 56: rc = act_bf16 ? launch_dw_cfg<28, 8, true>(w, C, x, y, B, H, W, s) : launch_dw_cfg<28, 8, false>(w, C, x, y, B, H, W, s); break;
            case 28: rc = act_bf16 ? launch_dw_cfg<28, 8, true>(w, C, x, y, B, H, W, s) : launch_dw_cfg<28, 8, false>(w, C, x, y, B, H, W, s); break;
            // stages 2 and 3: one row tile of one clip per workgroup (cache-resident tensors; the streaming kernel's long-lived
            // workgroups pay off only where every input row must come from HBM exactly once)
            case 14: rc = act_bf16 ? launch_dw_tile<14, true>(w, C, x, y, B, H, s) : launch_dw_tile<14, false>(w, C, x, y, B, H, s); break;
            case 7: rc = act_bf16 ? launch_dw_tile<7, true>(w, C, x, y, B, H, s) : launch_dw_tile<7, false>(w, C, x, y, B, H, s); break;
            default: ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: unsupported width %d (expected 56/28/14/7)", W);
        }
        ACX_TRY(rc);
    }
    if (stats) ACX_TRY(launch_rowstats(c, reinterpret_cast<const float*>(y), stats, (int64_t)B * H * W, C, s));
    return ACX_OK;
}

}  // namespace acx
