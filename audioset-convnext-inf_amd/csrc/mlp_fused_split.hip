// K4fs -- the fused block MLP (LayerNorm -> pwconv1 -> GELU -> pwconv2 -> gamma -> + residual, convnext.py:77-86)
// for stage 0 (C = 96) in split-fp16 arithmetic (ACX_PREC_F32_SPLIT, see gemm_split.hip for the number format and its
// error bound).  Each wave owns 32 pixels, everything is computed transposed so that the accumulator tile of the first
// product is the B operand of the second, the hidden activation never leaves the register file, and every fp32 operand
// is carried as fp16 hi + fp16 lo (three v_mfma_f32_32x32x16_f16 per product):
//     phase 1   X^T[32 hidden x 32 px] = W1c[32 x C] . LN(y)^T[C x 32 px]     3 C/16 MFMAs; B operand = the wave's
//               normalised activations (x 2^11) as hi/lo halves, resident in C/2 VGPRs: lane (px, h) holds
//               channels 16s + 8h .. +7 of k-step s
//     GELU      on the 16 accumulator registers (bias pre-loaded as the initial accumulator), x 2^h, split into
//               hi/lo halves -- SCALAR fp32 math, 10.5 VALU per element (split_math.h, gelu3_nano).  The file is built
//               with -fno-slp-vectorize: beside MFMAs a packed-FP32 instruction costs far more than the two scalar
//               ones it replaces (round 2 ran this kernel on v_pk_fma_f32 / v_pk_mul_f32: 104 of them per 36 MFMAs,
//               0.25 of the matrix peak; profiles/r03_a_coissue_table.txt prices one v_pk_fma_f32 in an MFMA gap at +18 cycles)
//     phase 2   out^T[C x 32 px] += W2c[C x 32 hidden] . G                      3 C/16 MFMAs.  Lane (px, h) holds hidden
//               units 4h + 8q + e (accumulator register 4q + e); registers 8s'..8s'+7 are used as-is for k-step s',
//               i.e. MFMA k-slot (s', h, j) contracts hidden unit 16 s' + 4h + 8 (j>>2) + (j&3) -- W2c is stored
//               with the hidden units of each chunk in exactly that order (acx_finalize), so its fragments are
//               plain 16-B reads.
// HBM traffic per block: read y, read x, write x.
//
// PERSISTENT workgroups (round 3).  A CU-exclusive workgroup cannot hide its own memory round trips behind a
// neighbour's arithmetic: in round 2 a 256-pixel workgroup lived ~40 us for ~17 us of loop -- the rest were the
// dependent HBM round trips of its prologue (y rows) and epilogue (x rows, stores) and the launch of the next workgroup.
// Now one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...; while it multiplies tile t, the y rows of its NEXT
// tile arrive by LDS-DMA (one 1-KB piece per wave and loop iteration, each lane fetching exactly the 16 bytes it will
// read back: a 96-KB buffer in lane order, conflict-free), the weight rings run on across the tile boundary, and the
// residual x rows are requested at the top of the last iteration into registers the GELU has just freed.
//
// Schedule (per wave and tile, iteration j over the n = 4C/32 = 12 hidden chunks) -- three instruction streams:
//     matrix  A: phase 1 of chunk j+2 (chained on one accumulator)       B: phase 2 of chunk j
//     vector  GELU + split of chunk j+1, cut into 64 micro-steps (8 register pairs x 8 steps of 2-6 instructions, two
//             pairs in flight so that consecutive micro-steps are independent) dealt one or two behind EACH MFMA of A
//             and B, behind scheduling fences
//     LDS     fragments of unit u+1 are read while unit u multiplies (two named register pairs)
//     DMA     waves 0-3 stream the W1c images (ring of 3: chunk j+4 requested in iteration j, wrapping into the next
//             tile's chunks 0-2), waves 4-7 the W2c images (ring of 2: chunk j+1 requested in iteration j), every wave one
//             piece of the next tile's y; the end-of-iteration wait is a counted s_waitcnt vmcnt(N) + bare s_barrier so
//             that the youngest W1c image stays in flight (a __syncthreads() would drain it: hipcc puts vmcnt(0) in
//             front of it)
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {

// Diagnostic builds only (-DACX_FS_STAMPS; tools/lab): s_memtime differences per loop section, summed per wave in scalar
// registers and written once at the end to a buffer nothing else reads.  The product build contains no stamp.
#ifdef ACX_FS_STAMPS
__device__ unsigned long long acx_fs_stamps[256 * 8 * 8];
#define ACX_STAMP_DECL unsigned long long st_prev_ = 0, st_sum_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ACX_STAMP(k_)                                                                                           \
    {                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        unsigned long long t_;                                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                            \
        if ((k_) >= 0) st_sum_[(k_) < 0 ? 0 : (k_)] += t_ - st_prev_;                                           \
        st_prev_ = t_;                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    }
#define ACX_STAMP_FLUSH                                                                                         \
    if (lane == 0 && blockIdx.x < 256) { _Pragma("unroll") for (int k = 0; k < 8; ++k) acx_fs_stamps[(blockIdx.x * 8 + wave) * 8 + k] = st_sum_[k]; }
#else
#define ACX_STAMP_DECL
#define ACX_STAMP(k_)
#define ACX_STAMP_FLUSH
#endif

template <int C>
struct FusedSCfg {
    // CU-exclusive workgroups (acx_internal.h): 8 waves, two per SIMD, 512 x 256 registers + all of the LDS
    static constexpr int kWaves = 8;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kPix = kWaves * 32;
    static constexpr int kLoaders = 4;                       // loader waves per weight image
    static constexpr int kChunks = 4 * C / 32;               // n
    static constexpr int kHalfBytes = 128 * C;               // one [32][C] (or [C][32]) S16 image
    static constexpr int kPieces = kHalfBytes / 1024 / kLoaders;    // 1-KB pieces per loader wave per image
    static constexpr int kRowChunks = C / 4;                 // 16-B chunks per W1c row
    static constexpr int kSteps = C / 16;                    // k-steps of phase 1
    static constexpr int kUnits = 2 * (C / 32);              // (out tile, k-step) units of phase 2
    static constexpr int kRing1 = 3, kRing2 = 2;             // W1c / W2c ring slots
    static constexpr int kYPieces = C / 8;                   // 16-B chunks of y per lane = 1-KB DMA pieces per wave and tile
    static constexpr int kYBytes = kWaves * kYPieces * 1024; // next tile's y rows, lane order
    static constexpr size_t kLdsBytes = (size_t)(kRing1 + kRing2) * kHalfBytes + kYBytes + 5 * C * 4;
    static_assert(kChunks % kRing1 == 0 && kChunks % kRing2 == 0, "ring slots must repeat from tile to tile");
    static_assert(kYPieces <= kChunks, "one y piece per iteration");
    __device__ static int swz1(int row) { return (C == 96) ? ((row >> 1) & 7) : (row & 15); }
};

template <int C, bool LNOUT>
__global__ __launch_bounds__(FusedSCfg<C>::kThreads) void mlp_fused_split_kernel(
    const float* __restrict__ y, float* __restrict__ x, const char* __restrict__ wpack /*[chunks][256*C bytes]*/,
    const float* __restrict__ b1, const float* __restrict__ b2, long long M, float sinv1, float sinv2, float hscale,
    char* __restrict__ ln_out /* LNOUT: (M, C) S16 rows of LayerNorm(x_new) x 2^11, written INSTEAD of x */,
    int ntiles) {
    using Cfg = FusedSCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int n = Cfg::kChunks, R1 = Cfg::kRing1, R2 = Cfg::kRing2;
    constexpr int kW2Off = R1 * Cfg::kHalfBytes;                       // [kRing2][kHalfBytes]  rows = out channel  (w1: [kRing1][kHalfBytes] at 0)
    constexpr int kYOff = (R1 + R2) * Cfg::kHalfBytes;                 // [waves][kYPieces][64 lanes][16 B]
    constexpr int kB1Off = kYOff + Cfg::kYBytes;                       // [4C] floats, pre-divided by sinv1; then b2: [C] floats
    char* const w1buf = smem;
    char* const w2buf = smem + kW2Off;
    float* const b1s = reinterpret_cast<float*>(smem + kB1Off);
    float* const b2s = b1s + 4 * C;
    // LDS byte addresses as plain integers from ONE address-space cast: every further cast costs a null check in scalar
    // code, and the loop below is bound by the number of instructions a wave issues, of whatever kind
    const unsigned smem_a = acx_lds_addr(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int lw = wave & 3;              // loader index inside its image
    const int role = wave >> 2;           // 0 = streams the W1c images, 1 = streams the W2c images
    ACX_CLAIM_VGPR(255);

    // ---- this wave's weight stream.  Requests advance cyclically by one chunk and one ring slot each:
    //   W1 waves: kernel start chunks 0 1 2 -> slots 0 1 2; per tile: chunk 3 after the tile prologue, chunks 4 .. 11, 0' 1' 2' in
    //             iterations 0 .. 10 (chunk j + 4 into slot (j + 1) % 3: the one chunk j + 1 left in iteration j - 1), none in 11
    //   W2 waves: kernel start chunk 0 -> slot 0; chunk j + 1 (0' after 11) in iteration j into slot (j + 1) % 2
    // State: wchunk = byte offset of the chunk block [W1c | W2c] to request next, wdst = LDS address of this wave's first piece
    // in the slot it goes to.  The three pieces of a request differ by 1 KB in LDS (instruction offset -- which also moves the
    // global address, so each piece's per-lane source offset is pre-biased by -1 KB x k) .
    // (round 5: scalar base + 32-bit lane offset, split_math.h acx_glds16_s -- the per-lane part of a piece's source is the swizzle only)
    unsigned wsrc[Cfg::kPieces];
#pragma unroll
    for (int k = 0; k < Cfg::kPieces; ++k) {
        const int idx = (lw * Cfg::kPieces + k) * 64 + lane;           // linear 16-B slot in the LDS image
        const int r1 = idx / Cfg::kRowChunks, p1 = idx - r1 * Cfg::kRowChunks;
        const int o1 = (r1 * Cfg::kRowChunks + (p1 ^ Cfg::swz1(r1))) * 16;
        const int r2 = idx >> 3, p2 = idx & 7;
        const int o2 = Cfg::kHalfBytes + (r2 * 8 + (p2 ^ ((r2 >> 1) & 7))) * 16;
        wsrc[k] = (unsigned)(role == 0 ? o1 : o2);
    }
    const unsigned wring_lo = smem_a + (role == 0 ? 0 : kW2Off) + lw * Cfg::kPieces * 1024;
    const unsigned wring_hi = wring_lo + (role == 0 ? R1 : R2) * Cfg::kHalfBytes;
    unsigned wdst = wring_lo;
    unsigned wchunk = 0;
#define ACX_WREQ_PIECE(k_) acx_glds16_s(wpack + wchunk, wsrc[k_], __builtin_amdgcn_readfirstlane(wdst + (k_) * 1024));
#define ACX_WREQ_ADVANCE                                                                                        \
        wchunk += 2 * Cfg::kHalfBytes; if (wchunk == (unsigned)n * 2 * Cfg::kHalfBytes) wchunk = 0;             \
        wdst += Cfg::kHalfBytes; if (wdst == wring_hi) wdst = wring_lo;
    // y piece q_ of the tile whose rows start at ynext: lane (px, hh) fetches the 16 bytes it will read back as its q-th chunk:
    // channels 16 (q >> 1) + 8 hh + 4 (q & 1) .. +3 of pixel row tile * kPix + wave * 32 + px (clamped into the tensor)
    const unsigned my_y_a = smem_a + kYOff + wave * (Cfg::kYPieces * 1024);
    // (scalar base = the tile's first row, 32-bit lane offset = this lane's row inside the tile: any tensor size)
    const float* ynext;       // wave-uniform
    unsigned yvoff;           // bytes from ynext to this lane's 8 hh-th channels of its (clamped) row
#define ACX_Y_ROWS(t_)                                                                                          \
        {                                                                                                       \
            long long t0_ = (long long)(t_) * Cfg::kPix;                                                        \
            if (t0_ >= M) t0_ = M - 1;                                                                          \
            long long r_ = (long long)(t_) * Cfg::kPix + wave * 32 + l31;                                       \
            if (r_ >= M) r_ = M - 1;                                                                            \
            ynext = y + t0_ * C;                                                                                \
            yvoff = (unsigned)(((r_ - t0_) * C + 8 * hh) * 4);                                                  \
        }
#define ACX_YREQ(q_) acx_glds16_s(ynext + (16 * ((q_) >> 1) + 4 * ((q_) & 1)), yvoff, __builtin_amdgcn_readfirstlane(my_y_a + (q_) * 1024));

    long long tile = blockIdx.x;              // the launcher keeps gridDim.x <= ntiles
    // ---- kernel prologue: the first tile's weights (W1c 0..2 / W2c 0) and y rows ---------------------------------------
#pragma unroll
    for (int c0 = 0; c0 < R1; ++c0) {
        if (c0 == 0 || role == 0) {
#pragma unroll
            for (int k = 0; k < Cfg::kPieces; ++k) ACX_WREQ_PIECE(k)
            ACX_WREQ_ADVANCE
        }
    }
    ACX_Y_ROWS(tile)
#pragma unroll
    for (int q = 0; q < Cfg::kYPieces; ++q) ACX_YREQ(q)
    {
        const float b1scale = 1.0f / sinv1;             // a power of two
        for (int i = tid; i < 4 * C; i += Cfg::kThreads) b1s[i] = b1[i] * b1scale;
        for (int i = tid; i < C; i += Cfg::kThreads) b2s[i] = b2[i];
    }

    const int sw1 = Cfg::swz1(l31);
    const int w1row = l31 * (4 * C);
    const int sw2 = (l31 >> 1) & 7;
    const int w2row = l31 * 128;
    GeluK3 gk = gelu_k3(sinv1, hscale);
    gelu_k3_to_vgprs(gk);
    constexpr int kGeluSteps = 8 * kGelu3Nano;

#define ACX_H8(v_) __builtin_bit_cast(h8, v_)
#define ACX_G4(g_, s_) __builtin_bit_cast(h8, uint4{g_[4 * (s_)], g_[4 * (s_) + 1], g_[4 * (s_) + 2], g_[4 * (s_) + 3]})
#define ACX_FENCE __builtin_amdgcn_sched_barrier(0);
    // phase-1 unit = k-step s_: two fragment reads, three MFMAs chained on Xacc
#define ACX_W1_RD(w1p_, s_, pl_) (*reinterpret_cast<const f32x4*>((w1p_) + w1row + (((2 * (2 * (s_) + hh) + (pl_)) ^ sw1) << 4)))
    // MFMA number m_ of an iteration (kTot of them) is followed, behind scheduling fences, by its share of the 64 GELU
    // micro-steps the iteration carries: steps [64 m / kTot, 64 (m + 1) / kTot)
#define ACX_AFTER(m_) ACX_FENCE if constexpr (HV) { ACX_MICRO(Xin, gnh, gnl, kGeluSteps * (m_) / kTot, kGeluSteps * ((m_) + 1) / kTot) } ACX_FENCE
#define ACX_P1_MFMA(Xacc_, s_, ah_, al_, m0_)                                                                   \
        Xacc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(acth[s_]), Xacc_, 0, 0, 0);          \
        ACX_AFTER((m0_) + 0)                                                                                    \
        Xacc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(actl[s_]), Xacc_, 0, 0, 0);          \
        ACX_AFTER((m0_) + 1)                                                                                    \
        Xacc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(acth[s_]), Xacc_, 0, 0, 0);          \
        ACX_AFTER((m0_) + 2)
#define ACX_P1_MFMA_PLAIN(Xacc_, s_, ah_, al_)                                                                  \
        Xacc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_H8(acth[s_]), Xacc_, 0, 0, 0);          \
        Xacc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(actl[s_]), Xacc_, 0, 0, 0);          \
        Xacc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_H8(acth[s_]), Xacc_, 0, 0, 0);
    // phase-2 unit i = (tile t = i >> 1, k-step s' = i & 1): fragments of block b = 2s' + hh, hi chunk 2b, lo chunk 2b+1
#define ACX_W2_RD(w2p_, i_, pl_) (*reinterpret_cast<const f32x4*>((w2p_) + ((i_) >> 1) * 4096 + (((2 * (2 * ((i_) & 1) + hh) + (pl_)) ^ sw2) << 4)))
#define ACX_P2_MFMA(i_, ah_, al_, m0_)                                                                          \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al_), ACX_G4(gch, (i_) & 1), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER((m0_) + 0)                                                                                    \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_G4(gcl, (i_) & 1), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER((m0_) + 1)                                                                                    \
        acc[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah_), ACX_G4(gch, (i_) & 1), acc[(i_) >> 1], 0, 0, 0); \
        ACX_AFTER((m0_) + 2)
    // both fragments of a unit behind ONE wait
#define ACX_TOUCH2(h_, l_) asm volatile("" :: "v"(h_), "v"(l_));
#define ACX_BIAS_INIT(Xacc_, j_)                                                                                \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * (j_) + 8 * q + 4 * hh);                 \
            Xacc_[4 * q + 0] = bq[0]; Xacc_[4 * q + 1] = bq[1]; Xacc_[4 * q + 2] = bq[2]; Xacc_[4 * q + 3] = bq[3]; \
        }
    // GELU steps [from, to) of the kGeluSteps = 8 x 21 single instructions that turn X_ into the packed halves gh_ / gl_ (8 words
    // each: the B operand of the next iteration's phase 2, k-step s' = words 4 s' .. 4 s' + 3): step m works on register pair
    // 2 (m / 42) + (m & 1) -- two pairs alternate, so neighbouring steps do not depend on each other -- and is step (m % 42) / 2
    // of that pair's twenty-one (split_math.h, gelu3_nano)
#define ACX_NANO_CASE(I_, X_, gh_, gl_) else if (st_ == (I_)) gelu3_nano<(I_)>(gs_, gk, X_[2 * pr_], X_[2 * pr_ + 1], gh_[pr_], gl_[pr_]);
#define ACX_MICRO(X_, gh_, gl_, from_, to_)                                                                     \
        _Pragma("unroll") for (int mm_ = (from_); mm_ < (to_); ++mm_) {                                         \
            const int pr_ = 2 * (mm_ / (2 * kGelu3Nano)) + (mm_ & 1), st_ = (mm_ % (2 * kGelu3Nano)) >> 1;      \
            GeluState3& gs_ = (mm_ & 1) ? gsB : gsA;                                                            \
            if (st_ == 0) gelu3_nano<0>(gs_, gk, X_[2 * pr_], X_[2 * pr_ + 1], gh_[pr_], gl_[pr_]);             \
            ACX_NANO_CASE(1, X_, gh_, gl_) ACX_NANO_CASE(2, X_, gh_, gl_) ACX_NANO_CASE(3, X_, gh_, gl_) ACX_NANO_CASE(4, X_, gh_, gl_) ACX_NANO_CASE(5, X_, gh_, gl_) ACX_NANO_CASE(6, X_, gh_, gl_) ACX_NANO_CASE(7, X_, gh_, gl_) \
            ACX_NANO_CASE(8, X_, gh_, gl_) ACX_NANO_CASE(9, X_, gh_, gl_) ACX_NANO_CASE(10, X_, gh_, gl_) ACX_NANO_CASE(11, X_, gh_, gl_) ACX_NANO_CASE(12, X_, gh_, gl_) ACX_NANO_CASE(13, X_, gh_, gl_) ACX_NANO_CASE(14, X_, gh_, gl_) \
            ACX_NANO_CASE(15, X_, gh_, gl_) ACX_NANO_CASE(16, X_, gh_, gl_) ACX_NANO_CASE(17, X_, gh_, gl_) ACX_NANO_CASE(18, X_, gh_, gl_) ACX_NANO_CASE(19, X_, gh_, gl_) ACX_NANO_CASE(20, X_, gh_, gl_) \
        }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // first tile: W1c(0..2), W2c(0), y landed ...
    __syncthreads();                                      // ... for every wave; b1s, b2s visible
    ACX_STAMP_DECL
    ACX_STAMP(-1)

    for (;;) {
        const long long next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        long long mrow = tile * Cfg::kPix + wave * 32 + l31;
        const bool valid = mrow < M;
        if (!valid) mrow = M - 1;
        ACX_Y_ROWS(next)      // (past the last tile: clamped rows, fetched for nothing -- cheaper than a branch in the loop)

        // ---- tile prologue: this wave's activations from the prefetch buffer; lane (px = l31, half hh) holds channels
        //      16s + 8hh .. +7, s = 0..C/16-1.  (Its own DMA pieces: covered by the waits of the previous tile's loop.)
        f32x4 acth[Cfg::kSteps], actl[Cfg::kSteps];         // 8 fp16 halves each
        {
            float a[C / 2];
            {
                // Read with inline asm: the pieces are this wave's own LDS-DMA writes, long landed (counted waits of the
                // previous tile's loop); hipcc knows nothing of them (they are issued from asm as well) and must not
                static_assert(Cfg::kYPieces == 12, "the asm below reads 12 pieces");
                f32x4 v[12];
                const unsigned ya = my_y_a + lane * 16;
                asm volatile(
                    "ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:1024\n\tds_read_b128 %2, %12 offset:2048\n\t"
                    "ds_read_b128 %3, %12 offset:3072\n\tds_read_b128 %4, %12 offset:4096\n\tds_read_b128 %5, %12 offset:5120\n\t"
                    "ds_read_b128 %6, %12 offset:6144\n\tds_read_b128 %7, %12 offset:7168\n\tds_read_b128 %8, %12 offset:8192\n\t"
                    "ds_read_b128 %9, %12 offset:9216\n\tds_read_b128 %10, %12 offset:10240\n\tds_read_b128 %11, %12 offset:11264\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                      "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
                    : "v"(ya) : "memory");
#pragma unroll
                for (int q = 0; q < Cfg::kYPieces; ++q) { a[4 * q + 0] = v[q][0]; a[4 * q + 1] = v[q][1]; a[4 * q + 2] = v[q][2]; a[4 * q + 3] = v[q][3]; }
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
                unsigned u4h[4], u4l[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    acx_split_pair((a[8 * s + 2 * p] - mean) * sc, (a[8 * s + 2 * p + 1] - mean) * sc, u4h[p], u4l[p]);
                }
                acth[s] = __builtin_bit_cast(f32x4, uint4{u4h[0], u4h[1], u4h[2], u4h[3]});
                actl[s] = __builtin_bit_cast(f32x4, uint4{u4l[0], u4l[1], u4l[2], u4l[3]});
            }
        }

        f32x16 acc[C / 32];
#pragma unroll
        for (int t = 0; t < C / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        // Two pre-activation tiles and two packed-G sets that swap roles from one iteration to the next (the loop is
        // unrolled by two): no register copies.  Iteration j reads Xin = X(j+1) (GELU input), accumulates Xacc = X(j+2), multiplies
        // with Gcur = G(j) and builds Gnew = G(j+1).
        f32x16 Xa, Xb;
        unsigned gah[8], gal[8], gbh[8], gbl[8];      // G as packed fp16 pairs: words 4 s' .. 4 s' + 3 = k-step s'
        GeluState3 gsA, gsB;
        f32x4 xr[C / 8];      // the residual x rows of the tile, requested in the last iteration
        {   // X(0) -> G(0) = Ga, X(1) -> Xa; nothing to overlap with yet
            ACX_BIAS_INIT(Xb, 0)
#pragma unroll
            for (int s = 0; s < Cfg::kSteps; ++s) {
                const f32x4 ah = ACX_W1_RD(w1buf, s, 0), al = ACX_W1_RD(w1buf, s, 1);
                ACX_P1_MFMA_PLAIN(Xb, s, ah, al)
            }
            ACX_MICRO(Xb, gah, gal, 0, kGeluSteps)
            ACX_BIAS_INIT(Xa, 1)
#pragma unroll
            for (int s = 0; s < Cfg::kSteps; ++s) {
                const f32x4 ah = ACX_W1_RD(w1buf + Cfg::kHalfBytes, s, 0), al = ACX_W1_RD(w1buf + Cfg::kHalfBytes, s, 1);
                ACX_P1_MFMA_PLAIN(Xa, s, ah, al)
            }
        }
        ACX_FENCE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // every wave is done with W1 ring slots 0 and 1 (and with its own y buffer)
        ACX_FENCE
        if (role == 0) {                   // W1c(3) -> slot 0; chunks 4.. follow in the iterations
#pragma unroll
            for (int k = 0; k < Cfg::kPieces; ++k) ACX_WREQ_PIECE(k)
            ACX_WREQ_ADVANCE
        }
        ACX_STAMP(0)                       // tile prologue

        auto body = [&](auto has_a, auto has_v, auto is_last, f32x16& Xin, f32x16& Xacc, unsigned (&gch)[8], unsigned (&gcl)[8],
                        unsigned (&gnh)[8], unsigned (&gnl)[8], const int j) __attribute__((always_inline)) {
            constexpr bool HA = decltype(has_a)::value, HV = decltype(has_v)::value, LAST = decltype(is_last)::value;
            constexpr int kRegions = (HA ? Cfg::kSteps : 0) + Cfg::kUnits;
            constexpr int kTot = 3 * kRegions;                     // MFMAs of this iteration
            // This iteration's requests, one per region from the start of the iteration (early: the W2c image has only
            // this iteration to land): region 0 the piece j of the next tile's y, regions 1 .. kPieces the wave's next
            // weight chunk -- unconditionally (a branch here would cut the iteration into basic blocks, and hipcc sinks the
            // GELU micro-steps out of their MFMA gaps across block boundaries), except that W1 waves skip the last iteration
            // (slot 0 keeps the next tile's chunk 0 until that tile's prologue has multiplied it)
#define ACX_DMA_AT(region_)                                                                                     \
            if ((region_) == 0) { ACX_YREQ(j) }                                                                 \
            else if ((region_) <= Cfg::kPieces) { if (!LAST || role == 1) { ACX_WREQ_PIECE((region_) - 1) } }
            ACX_FENCE
            if constexpr (LAST) {          // the residual: its registers were the GELU's until the previous iteration
                const float* xp = x + mrow * C + 4 * hh;
#pragma unroll
                for (int t = 0; t < C / 32; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xr[4 * t + q] = *reinterpret_cast<const f32x4*>(xp + 32 * t + 8 * q);
                ACX_FENCE
            }
            if constexpr (HA) {
                const char* w1p = w1buf + ((j + 2) % R1) * Cfg::kHalfBytes;
                ACX_BIAS_INIT(Xacc, j + 2)
                f32x4 a0h = ACX_W1_RD(w1p, 0, 0), a0l = ACX_W1_RD(w1p, 0, 1), a1h, a1l;
#pragma unroll
                for (int s = 0; s < Cfg::kSteps; s += 2) {
                    a1h = ACX_W1_RD(w1p, s + 1, 0); a1l = ACX_W1_RD(w1p, s + 1, 1);
                    ACX_FENCE
                    ACX_P1_MFMA(Xacc, s, a0h, a0l, 3 * (s))
                    ACX_DMA_AT(s)
                    ACX_FENCE
                    ACX_TOUCH2(a1h, a1l)
                    if (s + 2 < Cfg::kSteps) { a0h = ACX_W1_RD(w1p, s + 2, 0); a0l = ACX_W1_RD(w1p, s + 2, 1); }
                    ACX_FENCE
                    ACX_P1_MFMA(Xacc, s + 1, a1h, a1l, 3 * (s + 1))
                    ACX_DMA_AT(s + 1)
                    ACX_FENCE
                    if (s + 2 < Cfg::kSteps) ACX_TOUCH2(a0h, a0l)
                }
            }
            if constexpr (HA) { ACX_STAMP(1) }     // phase 1 + its share of the GELU
            {
                const char* w2p = w2buf + (j % R2) * Cfg::kHalfBytes + w2row;
                constexpr int r0 = HA ? Cfg::kSteps : 0;
                f32x4 a0h = ACX_W2_RD(w2p, 0, 0), a0l = ACX_W2_RD(w2p, 0, 1), a1h, a1l;
#pragma unroll
                for (int i = 0; i < Cfg::kUnits; i += 2) {
                    a1h = ACX_W2_RD(w2p, i + 1, 0); a1l = ACX_W2_RD(w2p, i + 1, 1);
                    ACX_FENCE
                    ACX_P2_MFMA(i, a0h, a0l, 3 * (r0 + i))
                    ACX_DMA_AT(r0 + i)
                    ACX_FENCE
                    ACX_TOUCH2(a1h, a1l)
                    if (i + 2 < Cfg::kUnits) { a0h = ACX_W2_RD(w2p, i + 2, 0); a0l = ACX_W2_RD(w2p, i + 2, 1); }
                    ACX_FENCE
                    ACX_P2_MFMA(i + 1, a1h, a1l, 3 * (r0 + i + 1))
                    ACX_DMA_AT(r0 + i + 1)
                    ACX_FENCE
                    if (i + 2 < Cfg::kUnits) ACX_TOUCH2(a0h, a0l)
                }
            }
#undef ACX_DMA_AT
            ACX_STAMP(HA ? 2 : 5)                  // phase 2 + its share of the GELU (5: the two short iterations)
            if (!LAST || role == 1) { ACX_WREQ_ADVANCE }
            ACX_FENCE
            ACX_STAMP(3)
            // W1 waves: the image requested this iteration may stay in flight, everything older (the y piece included)
            // must have landed; W2 waves: their image is multiplied in the next iteration -- all of it must have landed.
            // (vmcnt counts the stores of the previous tile's epilogue and the x loads of the last iteration too: in order.)
            if (role == 0 && !LAST) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::kPieces) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ACX_FENCE
            ACX_STAMP(4)                           // wait + barrier
        };
        static_assert(n % 2 == 0, "the iterations are unrolled in pairs");
        for (int j = 0; j + 2 < n; j += 2) {
            body(std::true_type{}, std::true_type{}, std::false_type{}, Xa, Xb, gah, gal, gbh, gbl, j);
            body(std::true_type{}, std::true_type{}, std::false_type{}, Xb, Xa, gbh, gbl, gah, gal, j + 1);
        }
        body(std::false_type{}, std::true_type{}, std::false_type{}, Xa, Xb, gah, gal, gbh, gbl, n - 2);
        body(std::false_type{}, std::false_type{}, std::true_type{}, Xb, Xa, gbh, gbl, gah, gal, n - 1);
        // The x rows have landed (vmcnt(0) above), but hipcc cannot see that wait: touch them here, on every path, so that its
        // own wait for them sits in front of the stores below -- left pending on the !valid path, they would be waited for
        // when their registers are rewritten at the top of the next tile, with vmcnt(N) that then waits for the stores too.
#pragma unroll
        for (int i = 0; i < C / 8; ++i) asm volatile("" : "+v"(xr[i]));

        // ---- tile epilogue: lane (px, hh), tile t, q: channels 32t + 8q + 4hh .. +3  ->  x = x + out + b2 ---------
        if constexpr (LNOUT) {
            // Last block of a stage in the full forward: the only reader of the new x is the LayerNorm in front of the
            // downsample conv (convnext.py:230-235 -- no skip connection leaves the stage), and this wave holds whole
            // rows (a lane and its partner at lane ^ 32 hold all C channels of one pixel).  So normalise here and
            // write the S16 operand of the downsample GEMM (same form as rowstats_kernel<.,3>: two-pass statistics,
            // biased variance, eps 1e-6, x 2^11, blocks [8 hi][8 lo]) in place of x: one tensor pass less on each
            // side and no LayerNorm launch.
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < C / 32; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * t + 8 * q;
                    const float4 bb = *reinterpret_cast<const float4*>(b2s + c + 4 * hh);
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
            // lanes (px, 0) and (px, 1) trade halves so that each writes ONE 16-byte piece per block -- the lower lane the 8 hi
            // halves, the upper lane the 8 lo halves: 32 contiguous bytes per row and store instruction (the rows are cold: 8-byte
            // pieces cost a read-for-ownership of every sector)
            char* op = ln_out + mrow * (long long)(C * 4) + 16 * hh;
#pragma unroll
            for (int t = 0; t < C / 32; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned uhi[2], ulo[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        acx_split_pair((acc[t][4 * q + 2 * e] - mean) * sc, (acc[t][4 * q + 2 * e + 1] - mean) * sc, uhi[e], ulo[e]);
                        acx_pair_swap(uhi[e], ulo[e]);
                    }
                    // block of channels 32t + 8q .. +7: [8 hi][8 lo]
                    if (valid) *reinterpret_cast<uint4*>(op + (4 * t + q) * 32) = uint4{uhi[0], uhi[1], ulo[0], ulo[1]};
                }
            }
        } else if (valid) {
            float* xp = x + mrow * C + 4 * hh;
#pragma unroll
            for (int t = 0; t < C / 32; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * t + 8 * q;
                    const float4 bb = *reinterpret_cast<const float4*>(b2s + c + 4 * hh);
                    f32x4 v = xr[4 * t + q];
                    v[0] += fmaf(acc[t][4 * q + 0], sinv2, bb.x);
                    v[1] += fmaf(acc[t][4 * q + 1], sinv2, bb.y);
                    v[2] += fmaf(acc[t][4 * q + 2], sinv2, bb.z);
                    v[3] += fmaf(acc[t][4 * q + 3], sinv2, bb.w);
                    *reinterpret_cast<f32x4*>(xp + c) = v;
                }
            }
        }
        ACX_STAMP(6)                               // tile epilogue
        if (!has_next) break;
        tile = next;
        ACX_FENCE
    }
    ACX_STAMP_FLUSH
#undef ACX_WREQ_PIECE
#undef ACX_WREQ_ADVANCE
#undef ACX_Y_ROWS
#undef ACX_YREQ
#undef ACX_H8
#undef ACX_G4
#undef ACX_FENCE
#undef ACX_W1_RD
#undef ACX_P1_MFMA
#undef ACX_P1_MFMA_PLAIN
#undef ACX_AFTER
#undef ACX_W2_RD
#undef ACX_P2_MFMA
#undef ACX_TOUCH2
#undef ACX_BIAS_INIT
#undef ACX_MICRO
}

template <int C, bool LNOUT>
static int launch_fused_s_cfg(const BlockW& w, const float* y, float* x, long long M, void* ln_out, hipStream_t s) {
    using Cfg = FusedSCfg<C>;
    static DeviceOnce once;
    static_assert(Cfg::kLdsBytes <= kCuLdsBytes, "weight rings + prefetch buffer do not fit the LDS");
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_split_kernel<C, LNOUT>, kCuLdsBytes));
    const long long ntiles = (M + Cfg::kPix - 1) / Cfg::kPix;
    if (ntiles > 0x7fffffffll) ACX_FAIL(ACX_ERR_SHAPE, "fused split MLP: %lld pixel tiles", ntiles);
    int cus = 0;
    ACX_TRY(cu_count_of_current_device(&cus));
    const long long blocks = ntiles < cus ? ntiles : cus;      // persistent: one workgroup per CU walks the tiles
    launch_kernel(&mlp_fused_split_kernel<C, LNOUT>, dim3((unsigned)blocks), dim3(Cfg::kThreads), kCuLdsBytes /* all of it: CU-exclusive */, s,
        y, x, reinterpret_cast<const char*>(w.wpack_s), w.b1, w.b2, M, 1.0f / (kSplitLnScale * w.w1s_scale),
        1.0f / (w.hid_scale * w.w2s_scale), w.hid_scale, reinterpret_cast<char*>(ln_out), (int)ntiles);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

bool mlp_fused_split_supported(int C) { return C == 96; }     // (C = 192, 384: mlp_fused_wide.hip)

int launch_mlp_fused_split(acx_ctx* c, const BlockW& w, int C, const float* y, float* x, long long M, hipStream_t s,
                           void* ln_out) {
    if (!w.wpack_s) ACX_FAIL(ACX_ERR_STATE, "fused split MLP: chunk-major S16 weights were not packed for C=%d", C);
    if (M <= 0) return ACX_OK;
    ProfScope ps(c, ACX_K_MLP_FUSED, s);
    if (C == 96) return ln_out ? launch_fused_s_cfg<96, true>(w, y, x, M, ln_out, s) : launch_fused_s_cfg<96, false>(w, y, x, M, nullptr, s);
    ACX_FAIL(ACX_ERR_SHAPE, "fused split MLP: unsupported channel count %d", C);
}

}  // namespace acx
