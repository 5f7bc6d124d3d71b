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
//   * the INPUT streams: group g = the rows g - 3 + q of the four lanes q (every LDS row is read by four consecutive groups), all
//     Q <= 6 input quads of the wave's strip.  A group meets kernel row g mod 4 of the output tile its rows start (X) and kernel row
//     g mod 4 + 4 of the tile above (Y), and is gone: registers hold the operands of ONE group, the raw pixels of the next, the
//     accumulators of two tiles of S <= 4 output quads (64) and the 21 x 2 weight operands (84) -- 220, two waves per SIMD, which
//     is what lets the packing of one wave run under the MFMAs of the other (one wave per SIMD with a 7-row window of packed
//     operands in registers, the first form, only matched the column kernel: profiles/r05_f_dwconv_mfma_v1_lab.txt);
//   * a GROUP of waves (4 / 2 / 1 for W = 56 / 28 / 14) splits an image row into strips of 4 4 3 3 / 4 3 / 4 output quads and
//     shares ONE ring of 16 whole rows of the slice: every wave requests a 1-KB piece of each row (LDS-DMA, counted s_waitcnt vmcnt),
//     one s_barrier per step says that everybody's pieces have landed and that the rows of the step before are free.  No strip re-reads
//     its neighbours' columns from the L2: private rings with halo quads moved 5/3 of the input and sat on the rate this access
//     pattern gets from the L2 (profiles/r05_f_halfline_bench.txt).  A row's pitch is NQ x 256 + 32 B: the four rows a 32-lane group
//     of a ds_read_b32 touches fall into different banks.  Rows between clips are ring rows the waves zero themselves (their
//     request re-reads an old row into a dummy row: the count stays fixed);
//   * a finished tile is packed to bf16 and leaves at the start of the NEXT step, behind that step's wait: stores sit in the same
//     in-order queue as the requests, and a wait must not stand behind stores that were issued a step ago;
//   * the weight operands are packed once per model in exactly the register form (api.hip, BlockW::dw_ops): 42 loads per wave;
//   * workgroups are numbered XCD-major and slices vary fastest: the two 64-byte halves of a 128-byte line and the six rows two
//     segments share meet in one L2.
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
constexpr int kDwmRing = 4 * kDwmD + 8;  // ring rows: the waves of a group read rows s - 3 .. s + 4 of a step while rows up to s + 4 D + 4 are requested

// S output quads; LEFT / RIGHT: the strip has an input quad left / right of its output quads (inside the image)
template <int W> struct DwmGeom {
    static constexpr int kC = 96 * 56 / W;
    static constexpr int kSlices = kC / 32;
    static constexpr int kNQ = (W + 3) / 4;                          // pixel quads of an image row (W = 14: the last one half outside)
    static constexpr int kStrips = W == 56 ? 4 : (W == 28 ? 2 : 1);  // waves across an image row: 4 4 3 3 / 4 3 / 4 output quads
    static constexpr int kGroups = 8 / kStrips;                      // (slice, segment) groups per 8-wave workgroup: the waves of a group share a ring
    static constexpr int kPitch = kNQ * 256 + 32;                    // ds_read_b32: 32 banks, lanes 0-31 = 4 rows x 8 channel pairs: rows 8 banks apart
    static constexpr int kGroupLds = (kDwmRing + 1) * kPitch;        // + 1: the dummy row
    static constexpr size_t kLdsBytes = (size_t)kGroups * kGroupLds;
    static constexpr int kGRowB = W * kC * 2;
    static_assert(kLdsBytes <= kCuLdsBytes, "the rings of a workgroup's groups must fit the LDS");
};
template <int W, int S, int LEFT, int RIGHT>
struct DwmCfg {
    static constexpr int kC = 96 * 56 / W;
    static constexpr int kQ = S + LEFT + RIGHT;
    static constexpr int kStores = 4 * S;
    static constexpr int kWait = (kDwmD - 1) * (kStores + 4);        // a step: its stores, then one request per row and wave
    static_assert(kWait <= 63, "vmcnt is a 6-bit counter");
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
__device__ __forceinline__ void dwm_run(const char* __restrict__ x, char* __restrict__ y, const char* __restrict__ ops,
                                        const float* __restrict__ bias, char* sink, char* lds, int piece, int B, int H, int steps, int seg,
                                        int slice, int jt0, unsigned magic, int item) {
    using Cfg = DwmCfg<W, S, LEFT, RIGHT>;
    using G = DwmGeom<W>;
    constexpr int C = Cfg::kC, Q = Cfg::kQ, kPitch = G::kPitch, R = kDwmRing;
    const int lane = threadIdx.x & 63, cl = lane >> 2, q = lane & 3;
    char* const ring = lds;                                // the group's ring: whole image rows of the slice, shared by the waves (strips) of the group
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    const int ch0 = slice * 32;
    const int col_in0 = 4 * (jt0 - LEFT);
    ACX_DWM_STAMP(0)

    // ---- the request side: the waves of a group fetch one 1-KB piece (16 pixels x 64 B) of every row each;
    // lane -> (pixel of the piece, 16-byte chunk of its 64-byte channel slice)
    const int rpx = piece * 16 + (lane >> 2);
    const bool vok = rpx < W;
    const unsigned voff = (unsigned)(((vok ? rpx : 0) * C + ch0) * 2 + (lane & 3) * 16);
    const unsigned poff = (unsigned)(piece * 1024);
    const int Lrows = 4 * steps;
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
        // (64-bit products are vector instructions even on uniform values: back to scalar registers once, here -- the cursors then
        // advance by scalar adds; without this every request re-read its pointer out of vector registers)
        qptr = dwm_scalar(x + ((long long)qn * H + (qr < H ? qr : H)) * G::kGRowB);
    }
    const char* safe_src = dwm_scalar(x + (long long)(qn < 0 ? 0 : (qn < B ? qn : B - 1)) * H * G::kGRowB);   // for the requests of non-image rows: a row of this wave's own neighbourhood
    // One request per call, always (the counted waits rely on it).  A row between clips: zeros written by the wave itself, the
    // request re-reads the wave's latest row into the dummy row; a row past the segment's last: the same without the zeros.
    auto request = [&]() {
        const bool need = qleft > 0;
        const bool real = need && qr < H && (unsigned)qn < (unsigned)B;
        const char* const src = real ? qptr : safe_src;
        const unsigned dst = ring_lds + poff + (real ? qoff : (unsigned)(R * kPitch));
        if (need && !real) {
            if (vok) *reinterpret_cast<dwm_f4*>(ring + qoff + poff + lane * 16) = dwm_f4{0.f, 0.f, 0.f, 0.f};
        }
        safe_src = src;
        qptr += real ? G::kGRowB : 0;
        qoff = qoff + kPitch == (unsigned)(R * kPitch) ? 0u : qoff + kPitch;
        const bool wrap = qr + 1 == Hp;
        qr = wrap ? 0 : qr + 1;
        qn += wrap ? 1 : 0;
        qleft -= 1;
        if (ACX_DWM_ABLATE == 2) return;
        if (vok) acx_glds16_s(dwm_scalar(src), voff, __builtin_amdgcn_readfirstlane(dst));
    };
    int on, orow;                                          // the output cursor
    char* optr;
    {
        const int n = (int)__umulhi((unsigned)vb, magic);
        on = n; orow = vb - n * Hp;
        optr = const_cast<char*>(dwm_scalar(y + ((long long)n * H + (orow < H ? orow : H)) * G::kGRowB));
    }

    // W = 14: columns 14 and 15 of every ring row are zeros no request ever writes
    if constexpr (W == 14) {
        for (int i = lane; i < R * 2 * 4; i += 64) {
            const int c = i & 3, px = 14 + ((i >> 2) & 1), row = i >> 3;
            *reinterpret_cast<dwm_f4*>(ring + row * kPitch + px * 64 + c * 16) = dwm_f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the first 4 D + 4 rows: vb - 3 .. vb + 4 D (everything the steps before the first in-loop request's target read)
#pragma unroll 1
    for (int i = 0; i < 4 * kDwmD + 4; ++i) {
        request();
#ifdef ACX_DWM_STAMPS_PROLOGUE      // lab: where the start of a wave goes -- stamps 5.. after each of the first requests
        ACX_DWM_STAMP(5 + i)
#endif
    }
    ACX_DWM_STAMP(1)
    // ---- weights: the lane's 21 x 2 operands B[kh][d = kq - jt + 1][set], packed once per model in exactly this form
    // (api.hip, dw_ops: [slice][kh][d][set][lane] x 4 bf16): 42 coalesced 8-byte loads, in flight beside the first rows
    dwm_s4 Bw[7][3][2];
    {
        const char* const ol = ops + (size_t)slice * (42 * 512) + lane * 8;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int st = 0; st < 2; ++st) Bw[kh][d][st] = *reinterpret_cast<const dwm_s4*>(ol + ((kh * 3 + d) * 2 + st) * 512);
    }
    const dwm_f2 bv = *reinterpret_cast<const dwm_f2*>(bias + ch0 + 2 * cl);
    ACX_DWM_STAMP(2)

    // ---- the data side: group g (lane q reads stacked row g - 3 + q), all Q quads, both channel sets.  A group meets kernel row
    // kh1 = g mod 4 of the tile its rows start (X) and kernel row kh1 + 4 of the tile above (Y): nothing but the current group's
    // operands and the two tiles' accumulators is kept in registers.
    unsigned raw[Q][4];
    int fidx = 0;                                          // ring slot of the next group's first row
    auto fetch = [&]() {
        unsigned t = (unsigned)fidx + (unsigned)q;
        t = min(t, t - (unsigned)R);                       // t >= R: wrap (unsigned: t - R is huge otherwise)
        const char* p = ring + t * kPitch + col_in0 * 64 + cl * 4;
#pragma unroll
        for (int kq = 0; kq < Q; ++kq)
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[kq][e] = *reinterpret_cast<const unsigned*>(p + (kq * 4 + e) * 64);
        fidx = fidx + 1 == R ? 0 : fidx + 1;
    };
    const unsigned lane_off = (unsigned)(((4 * jt0 + q) * C + ch0 + 2 * cl) * 2);
    const dwm_f4 b0 = {bv.x, bv.x, bv.x, bv.x}, b1 = {bv.y, bv.y, bv.y, bv.y};
    dwm_f4 acc[2][S][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jt = 0; jt < S; ++jt) { acc[t][jt][0] = b0; acc[t][jt][1] = b1; }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (G::kStrips > 1) __builtin_amdgcn_s_barrier();      // every wave's pieces of the first rows have landed (W = 14: a group is one wave, its ring its own)
    ACX_DWM_STAMP(3)
    fetch();
    ACX_DWM_STAMP(4)

#define ACX_DWM_KH(t_, kh_)                                                                                     \
    _Pragma("unroll") for (int jt_ = 0; jt_ < S; ++jt_)                                                         \
        _Pragma("unroll") for (int d_ = 0; d_ < 3; ++d_) {                                                      \
            const int kq_ = jt_ + d_ - 1 + LEFT;                                                                \
            if (kq_ < 0 || kq_ >= Q) continue;                                                                  \
            /* a tile's first instruction (kernel row 0, its first input quad) takes the bias as its C operand: no 32 register \
               moves per step to start the accumulators from */                                                 \
            const bool first_ = (kh_) == 0 && d_ == (jt_ - 1 + LEFT >= 0 ? 0 : 1);                              \
            acc[t_][jt_][0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(Ac[kq_][0], Bw[kh_][d_][0], first_ ? b0 : acc[t_][jt_][0], 0, 0, 0); \
            acc[t_][jt_][1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(Ac[kq_][1], Bw[kh_][d_][1], first_ ? b1 : acc[t_][jt_][1], 0, 0, 0); \
        }
    // group P of the eight-group period (P >> 2: which accumulator set is X)
#define ACX_DWM_GROUP(P_)                                                                                       \
    {                                                                                                           \
        dwm_s4 Ac[Q][2];                                                                                        \
        _Pragma("unroll") for (int kq_ = 0; kq_ < Q; ++kq_) {                                                   \
            Ac[kq_][0] = dwm_op(dwm_lo(raw[kq_][0], raw[kq_][1]), dwm_lo(raw[kq_][2], raw[kq_][3]));            \
            Ac[kq_][1] = dwm_op(dwm_hi(raw[kq_][0], raw[kq_][1]), dwm_hi(raw[kq_][2], raw[kq_][3]));            \
        }                                                                                                       \
        fetch();                                                                                                \
        ACX_DWM_KH(((P_) >> 2) & 1, (P_) & 3)                                                                   \
        if constexpr (((P_) & 3) < 3 && ACX_DWM_ABLATE != 6) { ACX_DWM_KH((((P_) >> 2) & 1) ^ 1, ((P_) & 3) + 4) } \
    }
    // a step = four groups; the tile above (Y) is complete after the third and leaves; its registers start the tile below
#define ACX_DWM_STEP(H_)                                                                                        \
    {                                                                                                           \
        /* this wave's pieces of the step's rows have landed; behind the barrier everybody's have, and everybody has read \
           the rows of the step before: their slots take the requests */                                        \
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ACX_DWM_ABLATE == 2 ? 0 : Cfg::kWait) : "memory");           \
        if constexpr (G::kStrips > 1) __builtin_amdgcn_s_barrier();                                             \
        ACX_DWM_STORE()                                                                                         \
        request(); request(); request(); request();                                                             \
        ACX_DWM_GROUP(4 * (H_) + 0) ACX_DWM_GROUP(4 * (H_) + 1) ACX_DWM_GROUP(4 * (H_) + 2) ACX_DWM_GROUP(4 * (H_) + 3) \
        _Pragma("unroll") for (int jt_ = 0; jt_ < S; ++jt_) {                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) outp[jt_][i_] = acx_pack_bf16x2(acc[(H_) ^ 1][jt_][0][i_], acc[(H_) ^ 1][jt_][1][i_]); \
        }                                                                                                       \
    }
    // The tile a step completes leaves at the START of the next step, behind that step's wait: a store sits in the same in-order
    // queue as the requests, and the wait of step s + 1 must not stand behind stores issued one step earlier (their
    // acknowledgement takes as long as a row takes to arrive: WAIT_INST was half of every wave's life in stage 0).
#define ACX_DWM_STORE()                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                      \
            const bool real_ = ACX_DWM_ABLATE != 5 && live > 1 && orow < H && on < B;                           \
            /* scalar base + 32-bit lane offset: the address costs no vector instruction; W = 14: the two lanes of the last \
               quad that lie outside the image are masked out of the store */                                   \
            char* const d_ = real_ ? optr : sink;                                                               \
            optr += real_ ? G::kGRowB : 0;                                                                      \
            { const bool wrap_ = orow + 1 == Hp; orow = live > 1 ? (wrap_ ? 0 : orow + 1) : orow; on += (live > 1 && wrap_) ? 1 : 0; } \
            _Pragma("unroll") for (int jt_ = 0; jt_ < S; ++jt_) {                                               \
                if (!(W == 14 && jt_ == 3 && q >= 2))                                                           \
                    *reinterpret_cast<unsigned*>(d_ + (size_t)(lane_off + (unsigned)(jt_ * 4 * C * 2))) = outp[jt_][i_]; \
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
    // steps + 1 steps: the last one is the three groups below the segment's last tile (its kernel rows 4-6), run as an
    // ordinary step (its requests find nothing left to need; what it adds to the tile below the segment is never stored) --
    // a second copy of the step for it, and a third for an odd count, put 70 KB of code into stage 0's kernel
    int left = steps + 1;
    [[maybe_unused]] int nstamp = 0;
#pragma unroll 1
    for (;;) {
        ACX_DWM_STEP(0)
        if (--left == 0) break;
        ACX_DWM_STEP(1)
        if (--left == 0) break;
#ifndef ACX_DWM_STAMPS_PROLOGUE
        ACX_DWM_STAMP(5 + nstamp) ++nstamp;
#endif
    }
    ACX_DWM_STORE()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ACX_DWM_STAMP(15)
#undef ACX_DWM_STEP
#undef ACX_DWM_STORE
#undef ACX_DWM_GROUP
#undef ACX_DWM_KH
}

// CU-exclusive like every kernel of the library that runs 16-bit MFMAs on live data beside LDS-DMA operand traffic (DESIGN.md 3b,
// ADVICE r05): ONE workgroup of eight waves per CU -- two waves per SIMD as before, 512 threads x 256 registers claimed, all of the
// LDS requested at launch -- so that no foreign wave (packed-FP32 code of another stream or process) is ever co-resident with it.
// tools/check_exclusive.py verifies the descriptor of the shipped kernel.
template <int W>
__global__ __launch_bounds__(512) void dwconv7_mfma_kernel(const void* __restrict__ x_, void* __restrict__ y_, const void* __restrict__ ops_ /* dw_ops */,
                                                              const float* __restrict__ bias, void* __restrict__ sink_, int B, int H,
                                                              int steps /* 4-row steps per segment */, int n_groups, unsigned magic) {
    using G = DwmGeom<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ACX_CLAIM_VGPR(255);
    // Consecutive workgroups go to different XCDs (round-robin over 8): logical block = the blockIdx-th in XCD-major order, so
    // that neighbours in the group order -- the other half of every 128-B line (the next slice), the segment below (six shared
    // rows) -- meet in ONE L2 instead of being fetched by two.
    const int nb = (int)gridDim.x, per = nb >> 3, rem = nb & 7, xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
    const int lblock = xcd * per + (xcd < rem ? xcd : rem) + k;
    // a group = (slice, segment), slice fastest; its kStrips waves split the image row (and the requests of every row) between them
    const int strip = wave % G::kStrips;
    const int group = min(lblock * G::kGroups + wave / G::kStrips, n_groups - 1);   // a group past the last repeats it (identical stores)
    const int slice = group % G::kSlices, seg = group / G::kSlices;
    const int item = group * G::kStrips + strip;
    const char* x = reinterpret_cast<const char*>(x_);
    const char* wt = reinterpret_cast<const char*>(ops_);
    char* y = reinterpret_cast<char*>(y_);
    char* sink = reinterpret_cast<char*>(sink_) + (size_t)(item % kDwSinkWindows) * kDwSinkWindowBytes;
    char* lds = smem + (wave / G::kStrips) * G::kGroupLds;
    if constexpr (W == 56) {
        if (strip == 0) dwm_run<56, 4, 0, 1>(x, y, wt, bias, sink, lds, 0, B, H, steps, seg, slice, 0, magic, item);
        else if (strip == 1) dwm_run<56, 4, 1, 1>(x, y, wt, bias, sink, lds, 1, B, H, steps, seg, slice, 4, magic, item);
        else if (strip == 2) dwm_run<56, 3, 1, 1>(x, y, wt, bias, sink, lds, 2, B, H, steps, seg, slice, 8, magic, item);
        else dwm_run<56, 3, 1, 0>(x, y, wt, bias, sink, lds, 3, B, H, steps, seg, slice, 11, magic, item);
    } else if constexpr (W == 28) {
        if (strip == 0) dwm_run<28, 4, 0, 1>(x, y, wt, bias, sink, lds, 0, B, H, steps, seg, slice, 0, magic, item);
        else dwm_run<28, 3, 1, 0>(x, y, wt, bias, sink, lds, 1, B, H, steps, seg, slice, 4, magic, item);
    } else {
        dwm_run<14, 4, 0, 0>(x, y, wt, bias, sink, lds, 0, B, H, steps, seg, slice, 0, magic, item);
    }
}

template <int W>
static int launch_dw_mfma_w(const void* x, void* y, const void* wt, const float* bias, void* sink, int B, int H, int target_waves, hipStream_t s) {
    using G = DwmGeom<W>;
    const long long Vt = (long long)B * (H + 3) - 3;
    if ((2 * Vt + 16ll * (H + 3) + 64) * (H + 3) >= 0xffffffffll)      // exactness of v / (H + 3) by multiply-high, up to the rows the last segment requests
        ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: batch too tall for one launch (%d clips of %d rows)", B, H);
    long long segs = target_waves / (G::kStrips * G::kSlices);
    if (segs < 1) segs = 1;
    long long rows = (Vt + segs - 1) / segs;
    rows = (rows + 3) / 4 * 4;
    if (rows < 8) rows = 8;
    const long long n_seg = (Vt + rows - 1) / rows;
    const int n_groups = (int)(n_seg * G::kSlices);
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &dwconv7_mfma_kernel<W>, kCuLdsBytes));
    launch_kernel(&dwconv7_mfma_kernel<W>, dim3((unsigned)((n_groups + G::kGroups - 1) / G::kGroups)), dim3(512), kCuLdsBytes /* CU-exclusive */, s,
        x, y, wt, bias, sink, B, H, (int)(rows / 4), n_groups, (unsigned)(0x100000000ull / (unsigned)(H + 3)) + 1u);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// bf16 activations only (stages 0-2 of set_precision("bf16a")); target_waves as in launch_dwconv_col
int launch_dwconv_mfma(const void* x, void* y, const void* wt /* BlockW::dw_ops */, const float* bias, void* sink, int B, int H, int W,
                       int target_waves, hipStream_t s) {
    switch (W) {
        case 56: return launch_dw_mfma_w<56>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 28: return launch_dw_mfma_w<28>(x, y, wt, bias, sink, B, H, target_waves, s);
        case 14: return launch_dw_mfma_w<14>(x, y, wt, bias, sink, B, H, target_waves, s);
        default: ACX_FAIL(ACX_ERR_SHAPE, "dwconv7 (matrix form): unsupported width %d (expected 56/28/14)", W);
    }
}

}  // namespace acx
