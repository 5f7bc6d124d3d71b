// K3m -- depthwise 7x7 convolution (Block.dwconv, convnext.py:58-60 called at :76) on the MATRIX pipe, for bf16 activations
// (set_precision("bf16a"), stages 0-2; round 5).
//
// With bf16 activations the column-streaming kernel (dwconv_col.hip) is bound by the issue rate of its 49 multiply-adds per
// output, and no 16-bit dot or packed instruction of gfx950 does two taps faster than one v_pk_fma_f32 does
// (profiles/r05_a_valu_rates.txt).  The matrix pipe does: v_mfma_f32_4x4x4_16b_bf16 multiplies SIXTEEN independent 4x4x4 blocks,
// one per group of four lanes -- 1024 multiply-adds in ~8 cycles against 128 in 4.5 (tools/lab/mfma4_rates.hip,
// profiles/r05_f_mfma4_rates.txt).  A block is one CHANNEL (a depthwise conv mixes none):
//     out[i][j] += sum_k A[i][k] B[k][j]
//       i = 0..3   four consecutive output ROWS r0 + i
//       j = 0..3   the four output pixels of one pixel quad jt (columns 4 jt + j)
//       k = 0..3   the four input pixels of one pixel quad kq = jt - 1, jt, jt + 1 (columns 4 kq + k), input row r0 + i + kh - 3
//       A[i][k] = x[r0 + i + kh - 3][4 kq + k]            B[k][j] = w[kh][4 (kq - jt) + k - j + 3]   (zero outside 0..6)
// so an output quad-row takes 7 kernel rows x 3 input quads = 21 instructions, 49 of whose 84 taps per output are real (58 % of
// the matrix rate; image edges drop the quads outside).  Weights are rounded to bf16 once (the activations already are),
// products are exact, accumulation is fp32 and starts from the bias.
//
// Layout of the work:
//   * a lane = (q = lane & 3: row i of A, column j of B and D;  cl = lane >> 2: the channel block).  A lane reads 4 B = channels
//     (2 cl, 2 cl + 1) of a pixel from the LDS; v_perm_b32 packs the low halves of two neighbouring pixels into an operand
//     register of channel 2 cl and the high halves into one of channel 2 cl + 1: a wave works on 32 channels as two channel sets;
//   * a WAVE owns a strip of S <= 4 output quads x 32 channels and walks DOWN a segment of the batch stacked as one tall image
//     (dwconv.hip: H rows of a clip, 3 rows of zeros, the next clip), four output rows per step.  Lane q holds the seven input
//     rows r0 + q - 3 .. r0 + q + 3 of the strip's Q <= 5 input quads as packed operands (8 row slots x Q quads x 2 sets x 2
//     registers); a step retires four of them and loads four (each LDS row is read by four lanes: x4 LDS reads, all the
//     re-use of a row across the 7 kernel rows and 3 output quads happens in registers);
//   * rows arrive by LDS-DMA into a ring of 16 rows PRIVATE to the wave (no barrier in the kernel, counted s_waitcnt vmcnt as in
//     dwconv_col.hip); a row's pitch is Q x 256 + 32 B, so the four rows a 32-lane group of a read touches fall into different banks;
//     rows between clips are ring rows the wave zeroes itself (their request goes to a dummy row, the count stays fixed);
//   * the 21 x 2 weight operands of a lane are built once per wave through a bf16 table in the LDS.
#include "acx_internal.h"
#include "split_math.h"

namespace acx {

typedef short dwm_s4 __attribute__((ext_vector_type(4)));
typedef float dwm_f4 __attribute__((ext_vector_type(4)));
typedef float dwm_f2 __attribute__((ext_vector_type(2)));
typedef unsigned dwm_u2 __attribute__((ext_vector_type(2)));

// lab builds only (tools/lab/dwm_lab.hip): 2 no DMA (the ring keeps stale bytes), 5 every store goes to the sink, 6 no MFMAs of the
// tile above (kernel rows 4-6) (timing, wrong results)
#ifndef ACX_DWM_ABLATE
#define ACX_DWM_ABLATE 0
#endif

// lab builds only: -DACX_DWM_STAMPS -> s_memtime at the marks of every wave, written to acx_dwm_stamps[item][16]
#ifdef ACX_DWM_STAMPS
__device__ unsigned long long acx_dwm_stamps[4096 * 16];
#define ACX_DWM_STAMP(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0 && item < 4096 && (k_) < 16) acx_dwm_stamps[item * 16 + (k_)] = t_; }
#else
#define ACX_DWM_STAMP(k_)
#endif

constexpr int kDwmD = 2;                 // steps (of four rows) a row is requested ahead of the step that first reads it
constexpr int kDwmRing = 4 * kDwmD + 4;  // ring rows: a step reads rows s - 3 .. s + 4 while rows up to s + 4 D + 4 are requested
constexpr int kDwmTabB = 7 * 16 * 64;    // weight table: [kernel row][tap + 4, 16 entries][32 channels] bf16

// S output quads; LEFT / RIGHT: the strip has an input quad left / right of its output quads (inside the image)
template <int W, int S, int LEFT, int RIGHT>
struct DwmCfg {
    static constexpr int kC = 96 * 56 / W;
    static constexpr int kQ = S + LEFT + RIGHT;
    static constexpr int kCols = W == 14 ? 14 : 4 * kQ;           // in-image input columns (W = 14: the last quad is half outside)
    static constexpr int kPieces = (kCols + 15) / 16;             // 1-KB DMA pieces per row: 16 pixels x 64 B
    static constexpr int kPitch = kQ * 256 + 32;           // ds_read_b32: 32 banks, lanes 0-31 = 4 rows x 8 channel pairs: rows 8 banks apart
    static constexpr int kStores = 4 * S;
    static constexpr int kWait = (kDwmD - 1) * (kStores + 4 * kPieces);
    static constexpr int kGRowB = W * kC * 2;
    static_assert(kWait <= 63, "vmcnt is a 6-bit counter");
};
template <int W> struct DwmGeom {
    static constexpr int kC = 96 * 56 / W;
    static constexpr int kSMax = W == 56 ? 3 : 4;
    static constexpr int kNQ = (W + 3) / 4;
    static constexpr int kStrips = (kNQ + kSMax - 1) / kSMax;       // 5 (3 3 3 3 2), 2 (4 3), 1 (4)
    static constexpr int kUnits = kStrips * (kC / 32);
    static constexpr int kQMax = W == 14 ? 4 : 5;                   // input quads of the widest strip
    static constexpr int kWaveLds = (kDwmRing + 1) * (kQMax * 256 + 32);      // + 1: the dummy row; the weight table (7 KB) lives in the ring's bytes before the first request
    static_assert(kWaveLds >= kDwmTabB, "the weight table must fit the ring");
    static constexpr size_t kLdsBytes = (size_t)4 * kWaveLds;
};

__device__ __forceinline__ unsigned dwm_lo(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }   // (a.lo, b.lo)
__device__ __forceinline__ unsigned dwm_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // (a.hi, b.hi)
// a pointer the compiler cannot prove wave-uniform (selects between cursors carried around a loop) -> scalar registers
__device__ __forceinline__ const char* dwm_scalar(const char* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ dwm_s4 dwm_op(unsigned r0, unsigned r1) { dwm_u2 v; v.x = r0; v.y = r1; return __builtin_bit_cast(dwm_s4, v); }

template <int W, int S, int LEFT, int RIGHT>
__device__ __forceinline__ void dwm_run(const char* __restrict__ x, char* __restrict__ y, const float* __restrict__ wt,
                                        const float* __restrict__ bias, char* sink, char* lds, int B, int H, int steps2, int seg,
                                        int slice, int jt0, unsigned magic, int item) {
    using Cfg = DwmCfg<W, S, LEFT, RIGHT>;
    constexpr int C = Cfg::kC, Q = Cfg::kQ, kPitch = Cfg::kPitch, R = kDwmRing;
    const int lane = threadIdx.x & 63, cl = lane >> 2, q = lane & 3;
    char* const ring = lds;
    char* const tab = lds;                                 // the weight table lives where the ring will: it is dead before the first request
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    const int ch0 = slice * 32;
    const int col_in0 = 4 * (jt0 - LEFT);
    ACX_DWM_STAMP(0)

    // ---- weights: bf16 table [kh][te = tap + 4][channel], then the lane's operands B[kh][d = kq - jt + 1][set]
    {
        // lane -> (channel pair cp, tap group tg): taps tg and tg + 4 of every kernel row, all 14 loads in flight at once (a loop
        // with a load and an LDS write per iteration paid a memory latency 28 times: half the life of a wave)
        const int cp = lane & 15, tg = lane >> 4;
        dwm_f2 v0[7], v1[7];
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            v0[kh] = *reinterpret_cast<const dwm_f2*>(wt + (kh * 7 + tg) * C + ch0 + 2 * cp);
            v1[kh] = *reinterpret_cast<const dwm_f2*>(wt + (kh * 7 + (tg < 3 ? tg + 4 : tg)) * C + ch0 + 2 * cp);
        }
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            char* const t = tab + kh * 1024 + cp * 4;
            *reinterpret_cast<unsigned*>(t + tg * 64) = 0u;
            *reinterpret_cast<unsigned*>(t + (4 + tg) * 64) = acx_pack_bf16x2(v0[kh].x, v0[kh].y);
            *reinterpret_cast<unsigned*>(t + (8 + tg) * 64) = tg < 3 ? acx_pack_bf16x2(v1[kh].x, v1[kh].y) : 0u;
            *reinterpret_cast<unsigned*>(t + (12 + tg) * 64) = 0u;
        }
    }
    const dwm_f2 bv = *reinterpret_cast<const dwm_f2*>(bias + ch0 + 2 * cl);
    dwm_s4 Bw[7][3][2];
    {
        const char* tl = tab + (3 - q) * 64 + cl * 4;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                unsigned L[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) L[k] = *reinterpret_cast<const unsigned*>(tl + kh * 1024 + (4 * d + k) * 64);
                Bw[kh][d][0] = dwm_op(dwm_lo(L[0], L[1]), dwm_lo(L[2], L[3]));
                Bw[kh][d][1] = dwm_op(dwm_hi(L[0], L[1]), dwm_hi(L[2], L[3]));
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the table has been read: its bytes are ring rows from here on
    ACX_DWM_STAMP(1)

    // ---- the request side: lane -> (pixel of the piece, 16-byte chunk of its 64-byte channel slice)
    unsigned voff[Cfg::kPieces];
    bool vok[Cfg::kPieces];
#pragma unroll
    for (int p = 0; p < Cfg::kPieces; ++p) {
        const int px = p * 16 + (lane >> 2);
        vok[p] = px < Cfg::kCols;
        voff[p] = (unsigned)(((col_in0 + (vok[p] ? px : 0)) * C + ch0) * 2 + (lane & 3) * 16);
    }
    const int Lrows = 8 * steps2;
    const int vb = seg * Lrows;                            // first output row (stacked) of the segment
    // Two cursors walk the stacked image row by row, in scalar registers (a division per row and request cost 360 cycles of
    // branches per request): clip n, row r of the clip (r >= H: one of the three rows between clips), pointer to the next image row.
    const int Hp = H + 3;
    int qn, qr, qleft = Lrows + 6;                         // the request cursor; rows the segment still needs: vb - 3 .. vb + L + 2
    unsigned qoff = 0;                                     // byte offset of its ring slot (row vb - 3 sits in slot 0)
    const char* qptr;
    {
        const int v0 = vb - 3 + Hp;                        // >= 0
        const int n1 = (int)__umulhi((unsigned)v0, magic);
        qr = v0 - n1 * Hp; qn = n1 - 1;
        qptr = x + ((long long)qn * H + (qr < H ? qr : H)) * Cfg::kGRowB;
    }
    const char* safe_src = x + (long long)(qn < 0 ? 0 : (qn < B ? qn : B - 1)) * H * Cfg::kGRowB;   // for the requests of non-image rows: a row of this wave's own neighbourhood
    // One request per call, always (the counted waits rely on it).  A row between clips: zeros written by the wave itself, the
    // request re-reads the wave's latest row into the dummy row; a row past the segment's last: the same without the zeros.
    auto request = [&]() {
        const bool need = qleft > 0;
        const bool real = need && qr < H && (unsigned)qn < (unsigned)B;
        const char* const src = real ? qptr : safe_src;
        const unsigned dst = ring_lds + (real ? qoff : (unsigned)(R * kPitch));
        if (need && !real) {
#pragma unroll
            for (int p = 0; p < Cfg::kPieces; ++p)
                if (vok[p]) *reinterpret_cast<dwm_f4*>(ring + qoff + p * 1024 + lane * 16) = dwm_f4{0.f, 0.f, 0.f, 0.f};
        }
        safe_src = src;
        qptr += real ? Cfg::kGRowB : 0;
        qoff = qoff + kPitch == (unsigned)(R * kPitch) ? 0u : qoff + kPitch;
        const bool wrap = qr + 1 == Hp;
        qr = wrap ? 0 : qr + 1;
        qn += wrap ? 1 : 0;
        qleft -= 1;
        if (ACX_DWM_ABLATE == 2) return;
#pragma unroll
        for (int p = 0; p < Cfg::kPieces; ++p)
            if (vok[p]) acx_glds16_s(dwm_scalar(src), voff[p], __builtin_amdgcn_readfirstlane(dst + p * 1024));
    };
    int on, orow;                                          // the output cursor
    char* optr;
    {
        const int n = (int)__umulhi((unsigned)vb, magic);
        on = n; orow = vb - n * Hp;
        optr = y + ((long long)n * H + (orow < H ? orow : H)) * Cfg::kGRowB;
    }

    // W = 14: columns 14 and 15 of every ring row are zeros no request ever writes
    if constexpr (Cfg::kCols < 4 * Q) {
        for (int i = lane; i < R * (4 * Q - Cfg::kCols) * 4; i += 64) {
            const int c = i & 3, px = Cfg::kCols + (i >> 2) % (4 * Q - Cfg::kCols), row = (i >> 2) / (4 * Q - Cfg::kCols);
            *reinterpret_cast<dwm_f4*>(ring + row * kPitch + px * 64 + c * 16) = dwm_f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the first R rows: vb - 3 .. vb + 4 D (everything the steps before the first in-loop request's target read)
#pragma unroll 1
    for (int i = 0; i < R; ++i) request();
    ACX_DWM_STAMP(2)

    // ---- the data side: group g (lane q reads stacked row g - 3 + q), all Q quads, both channel sets.  A group meets kernel row
    // kh1 = g mod 4 of the tile its rows start (X) and kernel row kh1 + 4 of the tile above (Y): nothing but the current group's
    // operands and the two tiles' accumulators is kept in registers.
    unsigned raw[Q][4];
    int fidx = 0;                                          // ring slot of the next group's first row
    auto fetch = [&]() {
        unsigned t = (unsigned)fidx + (unsigned)q;
        t = min(t, t - (unsigned)R);                       // t >= R: wrap (unsigned: t - R is huge otherwise)
        const char* p = ring + t * kPitch + cl * 4;
#pragma unroll
        for (int kq = 0; kq < Q; ++kq)
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[kq][e] = *reinterpret_cast<const unsigned*>(p + (kq * 4 + e) * 64);
        fidx = fidx + 1 == R ? 0 : fidx + 1;
    };
    const unsigned lane_off = (unsigned)(((4 * jt0 + q) * C + ch0 + 2 * cl) * 2);
    char* const slane = sink + lane_off;
    const dwm_f4 b0 = {bv.x, bv.x, bv.x, bv.x}, b1 = {bv.y, bv.y, bv.y, bv.y};
    dwm_f4 acc[2][S][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jt = 0; jt < S; ++jt) { acc[t][jt][0] = b0; acc[t][jt][1] = b1; }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ACX_DWM_STAMP(3)
    fetch();
    ACX_DWM_STAMP(4)

#define ACX_DWM_KH(t_, kh_)                                                                                     \
    _Pragma("unroll") for (int jt_ = 0; jt_ < S; ++jt_)                                                         \
        _Pragma("unroll") for (int d_ = 0; d_ < 3; ++d_) {                                                      \
            const int kq_ = jt_ + d_ - 1 + LEFT;                                                                \
            if (kq_ < 0 || kq_ >= Q) continue;                                                                  \
            acc[t_][jt_][0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(Ac[kq_][0], Bw[kh_][d_][0], acc[t_][jt_][0], 0, 0, 0); \
            acc[t_][jt_][1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(Ac[kq_][1], Bw[kh_][d_][1], acc[t_][jt_][1], 0, 0, 0); \
        }
    // group P of the eight-group period (P >> 2: which accumulator set is X); TAIL: the groups below the segment's last tile
#define ACX_DWM_GROUP(P_, TAIL_)                                                                                \
    {                                                                                                           \
        dwm_s4 Ac[Q][2];                                                                                        \
        _Pragma("unroll") for (int kq_ = 0; kq_ < Q; ++kq_) {                                                   \
            Ac[kq_][0] = dwm_op(dwm_lo(raw[kq_][0], raw[kq_][1]), dwm_lo(raw[kq_][2], raw[kq_][3]));            \
            Ac[kq_][1] = dwm_op(dwm_hi(raw[kq_][0], raw[kq_][1]), dwm_hi(raw[kq_][2], raw[kq_][3]));            \
        }                                                                                                       \
        fetch();                                                                                                \
        if (!(TAIL_)) { ACX_DWM_KH(((P_) >> 2) & 1, (P_) & 3) }                                                 \
        if (((P_) & 3) < 3 && ACX_DWM_ABLATE != 6) { ACX_DWM_KH((((P_) >> 2) & 1) ^ 1, ((P_) & 3) + 4) }        \
    }
    // a step = four groups; the tile above (Y) is complete after the third and leaves; its registers start the tile below
#define ACX_DWM_STEP(H_, TAIL_)                                                                                 \
    {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ACX_DWM_ABLATE == 2 ? 0 : Cfg::kWait) : "memory");           \
        ACX_DWM_STORE()                                                                                         \
        ACX_DWM_GROUP(4 * (H_) + 0, TAIL_) ACX_DWM_GROUP(4 * (H_) + 1, TAIL_) ACX_DWM_GROUP(4 * (H_) + 2, TAIL_) \
        if (!(TAIL_)) ACX_DWM_GROUP(4 * (H_) + 3, TAIL_)                                                        \
        _Pragma("unroll") for (int jt_ = 0; jt_ < S; ++jt_) {                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) outp[jt_][i_] = acx_pack_bf16x2(acc[(H_) ^ 1][jt_][0][i_], acc[(H_) ^ 1][jt_][1][i_]); \
            acc[(H_) ^ 1][jt_][0] = b0; acc[(H_) ^ 1][jt_][1] = b1;                                             \
        }                                                                                                       \
        if (!(TAIL_)) { request(); request(); request(); request(); }                                           \
    }
    // The tile a step completes leaves at the START of the next step, behind that step's wait: a store sits in the same in-order
    // queue as the requests, and the wait of step s + 1 must not stand behind stores issued one step earlier (their
    // acknowledgement takes as long as a row takes to arrive: WAIT_INST was half of every wave's life in stage 0).
#define ACX_DWM_STORE()                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                      \
            const bool real_ = ACX_DWM_ABLATE != 5 && live > 1 && orow < H && on < B;                           \
            char* const d_ = real_ ? optr + lane_off : slane;                                                   \
            optr += real_ ? Cfg::kGRowB : 0;                                                                    \
            { const bool wrap_ = orow + 1 == Hp; orow = live > 1 ? (wrap_ ? 0 : orow + 1) : orow; on += (live > 1 && wrap_) ? 1 : 0; } \
            _Pragma("unroll") for (int jt_ = 0; jt_ < S; ++jt_) {                                               \
                char* p_ = d_ + jt_ * 4 * C * 2;                                                                \
                if (W == 14 && LEFT + jt_ == 3 && q >= 2) p_ = slane + jt_ * 4 * C * 2;                         \
                *reinterpret_cast<unsigned*>(p_) = outp[jt_][i_];                                               \
            }                                                                                                   \
        }                                                                                                       \
        live += 1;                                                                                              \
    }
    unsigned outp[S][4];                                   // the finished tile, packed: stored by the next step
#pragma unroll
    for (int jt = 0; jt < S; ++jt)
#pragma unroll
        for (int i = 0; i < 4; ++i) outp[jt][i] = 0u;
    int live = 0;                                          // stores 0 and 1 carry nothing (no tile yet; the tile above the segment): sink
#pragma unroll 1
    for (int n2 = 0; n2 < steps2; ++n2) {
        ACX_DWM_STEP(0, false)
        ACX_DWM_STEP(1, false)
        ACX_DWM_STAMP(5 + n2)
    }
    ACX_DWM_STEP(0, true)
    ACX_DWM_STORE()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ACX_DWM_STAMP(15)
#undef ACX_DWM_STEP
#undef ACX_DWM_STORE
#undef ACX_DWM_GROUP
#undef ACX_DWM_KH
}

template <int W>
__global__ __launch_bounds__(256, 2) void dwconv7_mfma_kernel(const void* __restrict__ x_, void* __restrict__ y_, const float* __restrict__ wt /*[49][C]*/,
                                                           const float* __restrict__ bias, void* __restrict__ sink_, int B, int H,
                                                           int steps2 /* pairs of 4-row steps per segment */, int n_items, unsigned magic) {
    using G = DwmGeom<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Consecutive workgroups go to different XCDs (round-robin over 8): logical block = the blockIdx-th in XCD-major order, so
    // that neighbours in the item order -- the other half of every 128-B line (the next slice), the strip next door (a shared
    // quad column), the segment below (six shared rows) -- meet in ONE L2 instead of being fetched by two.
    const int nb = (int)gridDim.x, per = nb >> 3, rem = nb & 7, xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
    const int lblock = xcd * per + (xcd < rem ? xcd : rem) + k;
    const int item = min(lblock * 4 + wave, n_items - 1);              // a wave past the last item repeats it (identical stores)
    // slice fastest: a pixel's 64-byte slices (halves of 128-byte lines) are read by neighbouring waves of one workgroup
    const int unit = item % G::kUnits, seg = item / G::kUnits;
    const int slice = unit % (G::kC / 32), strip = unit / (G::kC / 32);
    const char* x = reinterpret_cast<const char*>(x_);
    char* y = reinterpret_cast<char*>(y_);
    char* sink = reinterpret_cast<char*>(sink_) + (size_t)(item % kDwSinkWindows) * kDwSinkWindowBytes;
    char* lds = smem + wave * G::kWaveLds;
    const int jt0 = strip * G::kSMax;
    if constexpr (W == 56) {
        if (strip == 0) dwm_run<56, 3, 0, 1>(x, y, wt, bias, sink, lds, B, H, steps2, seg, slice, jt0, magic, item);
        else if (strip == 4) dwm_run<56, 2, 1, 0>(x, y, wt, bias, sink, lds, B, H, steps2, seg, slice, jt0, magic, item);
        else dwm_run<56, 3, 1, 1>(x, y, wt, bias, sink, lds, B, H, steps2, seg, slice, jt0, magic, item);
    } else if constexpr (W == 28) {
        if (strip == 0) dwm_run<28, 4, 0, 1>(x, y, wt, bias, sink, lds, B, H, steps2, seg, slice, jt0, magic, item);
        else dwm_run<28, 3, 1, 0>(x, y, wt, bias, sink, lds, B, H, steps2, seg, slice, jt0, magic, item);
    } else {
        dwm_run<14, 4, 0, 0>(x, y, wt, bias, sink, lds, B, H, steps2, seg, slice, jt0, magic, item);
    }
}

template <int W>
static int launch_dw_mfma_w(const void* x, void* y, const float* wt, const float* bias, void* sink, int B, int H, int target_waves, hipStream_t s) {
    using G = DwmGeom<W>;
    const long long Vt = (long long)B * (H + 3) - 3;
    if ((2 * Vt + 16ll * (H + 3) + 64) * (H + 3) >= 0xffffffffll)      // exactness of v / (H + 3) by multiply-high, up to the rows the last segment requests
        ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: batch too tall for one launch (%d clips of %d rows)", B, H);
    long long segs = target_waves / G::kUnits;
    if (segs < 1) segs = 1;
    long long rows = (Vt + segs - 1) / segs;
    rows = (rows + 7) / 8 * 8;
    if (rows < 16) rows = 16;
    const long long n_seg = (Vt + rows - 1) / rows;
    const int n_items = (int)(n_seg * G::kUnits);
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &dwconv7_mfma_kernel<W>, G::kLdsBytes));
    launch_kernel(&dwconv7_mfma_kernel<W>, dim3((unsigned)((n_items + 3) / 4)), dim3(256), G::kLdsBytes, s,
        x, y, wt, bias, sink, B, H, (int)(rows / 8), n_items, (unsigned)(0x100000000ull / (unsigned)(H + 3)) + 1u);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// bf16 activations only (stages 0-2 of set_precision("bf16a")); target_waves as in launch_dwconv_col
int launch_dwconv_mfma(const void* x, void* y, const float* wt, const float* bias, void* sink, int B, int H, int W,
                       int target_waves, hipStream_t s) {
    switch (W) {
        case 56: return launch_dw_mfma_w<56>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 28: return launch_dw_mfma_w<28>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 14: return launch_dw_mfma_w<14>(x, y, wt, bias, sink, B, H, target_waves, s);
        default: ACX_FAIL(ACX_ERR_SHAPE, "dwconv7 (matrix form): unsupported width %d (expected 56/28/14)", W);
    }
}

}  // namespace acx
