// K4s/K5s -- the dense contractions with fp32 operands carried as TWO fp16 halves (ACX_PREC_F32_SPLIT).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the fp16 matrix rate on gfx950 (157 vs 2500 TFLOP/s).  An fp32 value v
// is v = hi + lo + e with hi = fp16(v), lo = fp16(v - hi), |e| <= 2^-23 |v| in the worst case (two round-to-nearest 11-bit
// pieces: exact for ~40 % of random fp32 mantissas, one bit short of fp32's 24 in the worst case -- measured on 2^20 values in
// tests/test_split_arithmetic_cpu.py; gfx950 MFMA honours fp16 subnormals -- tools/mfma_denorm_probe.hip -- and a
// power-of-two pre-scale keeps the pieces far from the bottom of the fp16 range).  A product of two such values is
//   a b = ah bh + ah bl + al bh + (al bl <= 2^-22 a b, dropped),
// every partial product of two fp16 numbers is exact in fp32 and the matrix core accumulates them in fp32, so three
// fp16 MFMAs reproduce an fp32 FMA chain to within the rounding of the fp32 accumulation itself -- at 16/3 of the
// f32-MFMA rate.  tests/test_gpu_parity.py runs its whole suite against this mode at the fp32 tolerances.
//
// Operand format "S16" (the same 4 bytes per element as fp32): a row of K values is K/8 blocks of 32 B,
//   [8 x fp16 hi][8 x fp16 lo];  a 128-B LDS row = 4 blocks = 32 k.  Lane half h of a 32x32x16 MFMA needs 8
// consecutive k of one row = one block: hi chunk 4s+2h, lo chunk 4s+2h+1 of the row (s = k-step of 16 in the tile).
// Producers: the LayerNorm pass (dwconv.hip, rows scaled by 2^11: |LN(y)| <= sqrt(C-1) < 28), this kernel's
// GELU epilogue (hidden activation scaled by 2^4, clamped to the fp16 range) and acx_finalize for the weights
// (scaled per layer to max |w| in [2^14, 2^15)).  The epilogue multiplies the accumulator by the exact inverse.
//
// Tiling / staging as gemm.hip, but one 8-wave workgroup per CU: 256 x BN x 32 per workgroup (4 x 2 waves, each 64 x BN/2),
// both operands by LDS-DMA with the XOR swizzle on the source address, fragments double-buffered in registers.
// The workgroup is CU-EXCLUSIVE (acx_internal.h, kCuLdsBytes): it claims all 160 KB of LDS and 512 x 256 registers, so no
// foreign wave can be co-resident with its MFMA loop (packed-FP32 VALU work of a co-resident wave goes wrong next to
// it on this platform -- tools/race2/).  A k-tile is only 2 x TM*TN*3 MFMAs of 32 cycles, so
// the pipeline is deeper than the fp32 kernel's: B tile t+2 and A tiles t+2, t+3 are in flight while tile t is multiplied.
#include <cstdlib>

#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {


constexpr int kSRowBytes = 128;     // 32 k per LDS row
constexpr int kSBK = 32;


struct GemmSParams {
    const char* A; const char* Wt; const float* bias; void* out; const float* resid;
    long long M; int N; int K;
    float sinv;                // 1 / (scale of A * scale of Wt)
    float hscale;              // EPI 1: scale of the S16 result (power of two)
    int H, W, C, Ho, Wo;       // gather mode: A is (B,H,W,C) S16 rows; row m = (b,ho,wo), k = (dy*2+dx)*C + c
    int tiles_n;
};

__device__ __forceinline__ void lds_dma16_s(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// EPI: 0 bias -> fp32, 1 bias + GELU -> S16 (scaled by kHiddenScale), 2 bias + residual -> fp32
template <int kBM, int BN, int WM, int WN, int EPI, int GATHER>
__global__ __launch_bounds__(64 * WM * WN) void gemm_split_kernel(GemmSParams p) {
    constexpr int TM = kBM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int NW = WM * WN;                                           // waves per workgroup (4 or 8)
    constexpr int A_TILE = kBM * kSRowBytes, B_TILE = BN * kSRowBytes;
    constexpr int A_DMA = kBM / (8 * NW), B_DMA = BN / (8 * NW);          // 1-KB pieces per wave per tile
    static_assert(A_DMA * 8 * NW == kBM && B_DMA * 8 * NW == BN, "tile rows must split into 8-row pieces per wave");
    // D = (W A^T): lane = row m, registers = 4 consecutive n -> 16-B accesses per lane in every epilogue
    // (the un-swapped form, 4-B accesses with n on the lanes, spent 73 of 289 us in pwconv2's read-modify-write)
    constexpr bool SWAP = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + 3 * A_TILE;        // A: ring of three k-tiles, B: two (see the k loop)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // claim the whole register budget of two waves per SIMD (with 512 threads hipcc keeps everything, accumulators
    // included, in v0..v255; tools/check_exclusive.py verifies the emitted descriptor)
    static_assert(NW == 8, "CU-exclusive launch: 8 waves x 256 registers");
    ACX_CLAIM_VGPR(255);
    long long lid = blockIdx.x;
    {   // XCD-contiguous tile order (see gemm.hip)
        const long long nwg = gridDim.x, per = (nwg + 7) >> 3, full = nwg - (per - 1) * 8;
        const long long xcd = lid & 7, k = lid >> 3;
        lid = (xcd < full ? xcd * per : full * per + (xcd - full) * (per - 1)) + k;
    }
    const int tile_n = (int)(lid % p.tiles_n);
    const long long tile_m = lid / p.tiles_n;
    const long long m0 = tile_m * kBM;
    const int n0 = tile_n * BN;

    const int prow = lane >> 3, pchunk = lane & 7;
    const char* a_src[A_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int row = A_DMA * 8 * wave + 8 * i + prow;
        const int chunk = pchunk ^ acx_swz8(row);
        long long m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        if (GATHER) {
            const int wo = (int)(m % p.Wo);
            const long long t = m / p.Wo;
            const int ho = (int)(t % p.Ho);
            const long long b = t / p.Ho;
            a_src[i] = p.A + (((b * p.H + 2 * ho) * p.W + 2 * wo) * p.C) * 4 + 16 * chunk;
        } else {
            a_src[i] = p.A + m * p.K * 4 + 16 * chunk;
        }
    }
    const char* b_src[B_DMA];
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int row = B_DMA * 8 * wave + 8 * i + prow;
        const int chunk = pchunk ^ acx_swz8(row);
        b_src[i] = p.Wt + (long long)(n0 + row) * p.K * 4 + 16 * chunk;
    }
    char* a_dst = As + A_DMA * 8 * wave * kSRowBytes;
    char* b_dst = Bs + B_DMA * 8 * wave * kSRowBytes;
    auto a_koff = [&](int k0) -> long long {        // byte offset of k-tile k0 inside an A row
        if (GATHER) {
            const int qd = k0 / p.C;
            return ((long long)((qd >> 1) * p.W + (qd & 1)) * p.C + (k0 - qd * p.C)) * 4;
        }
        return (long long)k0 * 4;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = acx_swz8(l31);
    int foff_hi[2], foff_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        foff_hi[s] = l31 * kSRowBytes + (((4 * s + 2 * hh) ^ sw) << 4);
        foff_lo[s] = l31 * kSRowBytes + (((4 * s + 2 * hh + 1) ^ sw) << 4);
    }
    const int a_frag_off = wm * TM * 32 * kSRowBytes;
    const int b_frag_off = wn * TN * 32 * kSRowBytes;
#define ACX_READ_FRAGS(F, abase, bbase, s)                                                              \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                               \
            F##ah[i] = *reinterpret_cast<const f32x4*>((abase) + i * 32 * kSRowBytes + foff_hi[s]);    \
            F##al[i] = *reinterpret_cast<const f32x4*>((abase) + i * 32 * kSRowBytes + foff_lo[s]);    \
        }                                                                                              \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                               \
            F##bh[j] = *reinterpret_cast<const f32x4*>((bbase) + j * 32 * kSRowBytes + foff_hi[s]);    \
            F##bl[j] = *reinterpret_cast<const f32x4*>((bbase) + j * 32 * kSRowBytes + foff_lo[s]);    \
        }                                                                                              \
    }
#define ACX_H8(x) __builtin_bit_cast(h8, x)
#define ACX_MFMA1(term, i, j, F)     /* term 0: lo x hi, 1: hi x lo, 2: hi x hi */                      \
    if (SWAP) {                                                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8((term) == 0 ? F##bl[j] : F##bh[j]),  \
                                                           ACX_H8((term) == 1 ? F##al[i] : F##ah[i]), acc[i][j], 0, 0, 0); \
    } else {                                                                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8((term) == 0 ? F##al[i] : F##ah[i]),  \
                                                           ACX_H8((term) == 1 ? F##bl[j] : F##bh[j]), acc[i][j], 0, 0, 0); \
    }
    // term-major order: MFMAs on the same accumulator are TM*TN instructions apart
#define ACX_PRIO_HI
#define ACX_PRIO_LO
#define ACX_MFMA_STEP(F)                                                                               \
    {                                                                                                  \
        ACX_PRIO_HI                                                                                    \
        _Pragma("unroll") for (int term = 0; term < 3; ++term)                                         \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) { ACX_MFMA1(term, i, j, F) }                    \
        ACX_PRIO_LO                                                                                    \
    }
    // the same with LDS-DMA pieces threaded in, one piece per MFMA in program order (hipcc may move a piece by an MFMA or
    // two; scheduling fences around every pair cost 5 %): first the B pieces of tile t+2, then -- behind one fence -- the A
    // pieces of tile t+3 (the counted wait at the next barrier relies on this order, see the k loop)
#define ACX_MFMA_STEP_DMA(F, koffA, aslot, k0B, bslot)                                                 \
    {                                                                                                  \
        static_assert(A_DMA + B_DMA <= 3 * TM * TN, "one DMA piece per MFMA");                         \
        _Pragma("unroll") for (int term = 0; term < 3; ++term)                                         \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                               \
            const int pc = (term * TM + i) * TN + j;                                                   \
            if (pc == B_DMA) __builtin_amdgcn_sched_barrier(0);      /* every B piece is issued before the first A piece */ \
            if (pc < B_DMA) lds_dma16_s(b_src[pc] + (k0B), b_dst + (bslot) * B_TILE + pc * 8 * kSRowBytes); \
            else if (pc < A_DMA + B_DMA)                                                               \
                lds_dma16_s(a_src[pc - B_DMA] + (koffA), a_dst + (aslot) * A_TILE + (pc - B_DMA) * 8 * kSRowBytes); \
            ACX_MFMA1(term, i, j, F)                                                                   \
        }                                                                                              \
    }
#define ACX_DMA_A(koffA, aslot)                                                                        \
    {   _Pragma("unroll") for (int i = 0; i < A_DMA; ++i)                                              \
            lds_dma16_s(a_src[i] + (koffA), a_dst + (aslot) * A_TILE + i * 8 * kSRowBytes); }
#define ACX_DMA_B(k0B, bslot)                                                                          \
    {   _Pragma("unroll") for (int i = 0; i < B_DMA; ++i)                                              \
            lds_dma16_s(b_src[i] + (k0B), b_dst + (bslot) * B_TILE + i * 8 * kSRowBytes); }
#define ACX_TOUCH(F)      /* see gemm.hip: keeps hipcc's lgkmcnt(0) off freshly issued reads */        \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) { asm volatile("" :: "v"(F##ah[i])); asm volatile("" :: "v"(F##al[i])); } \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) { asm volatile("" :: "v"(F##bh[j])); asm volatile("" :: "v"(F##bl[j])); } \
    }

    const int nk = p.K / kSBK;
    auto ktile = [&](int t) { return (t < nk ? t : nk - 1) * kSBK; };      // past the end: re-request the last tile (harmless, keeps the loop uniform)
    // prologue: tile 0 landed; then B(1), A(1), A(2) in flight -- in THIS order -- and the fragments of tile 0 in registers
    ACX_DMA_A(a_koff(0), 0)
    ACX_DMA_B(0LL, 0)
    __syncthreads();
    ACX_DMA_B((long long)ktile(1) * 4, 1)                      // nk >= 2 (checked by the launcher)
    ACX_DMA_A(a_koff(ktile(1)), 1)
    ACX_DMA_A(a_koff(ktile(2)), 2)
    f32x4 F0ah[TM], F0al[TM], F0bh[TN], F0bl[TN], F1ah[TM], F1al[TM], F1bh[TN], F1bl[TN];
    {
        const char* ab = As + a_frag_off;
        const char* bb = Bs + b_frag_off;
        ACX_READ_FRAGS(F0, ab, bb, 0)
        ACX_READ_FRAGS(F1, ab, bb, 1)
    }
    // steady state, tile t (A slot t % 3, B slot t & 1):
    //   MFMA s0 | counted wait + barrier (tile t+1 landed, tile t fully read by every wave) | rd s0(t+1) |
    //   MFMA s1 threaded with DMA B(t+2) -> B slot of t, then DMA A(t+3) -> A slot of t | rd s1(t+1)
    // The activations (A) come from HBM / the Infinity Cache: with a 2-slot ring a request had ONE k-tile (~1 us) to land
    // and the loop spent half its time in the barrier's wait; the third A slot gives it two.  One wave's vector-memory
    // counter retires in issue order, so "tile t+1 landed" = everything but the newest A_DMA pieces (those of A(t+2),
    // issued last in the previous iteration) has returned: s_waitcnt vmcnt(A_DMA).
    // (no conditional inside the loop: a branch around the DMA variant makes hipcc copy all accumulators twice per
    //  iteration; past the last tile the requests repeat tile nk-1 into slots nobody reads again)
    int a_cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int a_nxt = a_cur == 2 ? 0 : a_cur + 1;
        const char* abn = As + a_nxt * A_TILE + a_frag_off;
        const char* bbn = Bs + ((kt + 1) & 1) * B_TILE + b_frag_off;
        const long long kb = (long long)ktile(kt + 2) * 4;
        const long long ka = a_koff(ktile(kt + 3));
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_STEP(F0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(F1)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_DMA) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ACX_READ_FRAGS(F0, abn, bbn, 0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_STEP_DMA(F1, ka, a_cur, kb, kt & 1)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(F0)
        ACX_READ_FRAGS(F1, abn, bbn, 1)
        a_cur = a_nxt;
    }
    ACX_MFMA_STEP(F0)
    ACX_MFMA_STEP(F1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the repeated requests of the last iterations: nothing may still be writing the LDS when the workgroup ends
#undef ACX_READ_FRAGS
#undef ACX_MFMA1
#undef ACX_MFMA_STEP
#undef ACX_MFMA_STEP_DMA
#undef ACX_DMA_A
#undef ACX_DMA_B
#undef ACX_TOUCH
#undef ACX_H8

    const float sinv = p.sinv;
    f32x4 bjq[TN][4];             // (round 4: biases loaded once, not once per store -- see gemm_split16_kernel's epilogue)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bjq[j][q] = *reinterpret_cast<const f32x4*>(p.bias + n0 + (wn * TN + j) * 32 + 8 * q + 4 * hh);
    if (EPI == 1) {
        const GeluK3 gk = gelu_k3(sinv, p.hscale);      // GELU of v = a * sinv, result x p.hscale (split_math.h, third form)
        const float binv = 1.0f / sinv;     // a power of two
        // ---- GELU epilogue, D = W A^T: lane = row m, registers r = 4q+e hold n = 8q + 4hh + e ------------------
        // One S16 block (8 n) = [hi x8][lo x8] is shared by the lane pair (l31, hh=0/1): after a permlane32 swap
        // the low lane holds all 8 hi halves and the high lane all 8 lo halves -> one 16-B store each.
        char* outb = reinterpret_cast<char*>(p.out);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long m = m0 + (wm * TM + i) * 32 + l31;
            const bool ok = m < p.M;
            char* orow = outb + (ok ? m : 0) * p.N * 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = n0 + (wn * TN + j) * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = bjq[j][q];
                    unsigned xh[2], xl[2];
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {      // acc holds v / sinv - bias / sinv: add the pre-scaled bias first
                        const float ax = acc[i][j][4 * q + 2 * e2] + b4[2 * e2] * binv;
                        const float ay = acc[i][j][4 * q + 2 * e2 + 1] + b4[2 * e2 + 1] * binv;
                        gelu3_pair(gk, ax, ay, xh[e2], xl[e2]);
                    }
                    // low lanes: (own hi, partner hi); high lanes: (partner lo, own lo)
                    auto r0 = __builtin_amdgcn_permlane32_swap(xh[0], xl[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(xh[1], xl[1], false, false);
                    uint4 o;
                    o.x = r0[0]; o.y = r1[0]; o.z = r0[1]; o.w = r1[1];
                    if (ok) *reinterpret_cast<uint4*>(orow + (long long)(nb + 8 * q) * 4 + 16 * hh) = o;
                }
            }
        }
    } else {
        // ---- fp32 epilogue, same layout: lane (l31, hh) owns row m, columns nb + 8q + 4hh .. +3 of tile (i, j) ------
        float* outf = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long m = m0 + (wm * TM + i) * 32 + l31;
            const bool ok = m < p.M;
            const long long row = (ok ? m : 0) * p.N;
            f32x4 rv[TN][4];
            if (EPI == 2) {       // all residual loads of the row tile in flight before the first store
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        rv[j][q] = *reinterpret_cast<const f32x4*>(p.resid + row + n0 + (wn * TN + j) * 32 + 8 * q + 4 * hh);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = n0 + (wn * TN + j) * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = bjq[j][q];
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaf(acc[i][j][4 * q + e], sinv, b4[e]);
                        if (EPI == 2) v[e] += rv[j][q][e];
                    }
                    if (ok) *reinterpret_cast<f32x4*>(outf + row + nb + 8 * q + 4 * hh) = v;
                }
            }
        }
    }
}

// The same GEMM on v_mfma_f32_16x16x32_f16 (round 3).  Under MFMA load the chip holds a higher clock on this shape than
// on 32x32x16 at equal cycles per flop (MI355X_MICROARCH.md 'DVFS give-back' item 7; profiles/r03_i_mfma_shape_rates.txt:
// bare loops on random operands +5..10 % TFLOP/s), and a timing-only substitution in this kernel took 5-7 % off its
// launches.  Same tiles (256 x 192 x 32, 4 x 2 waves of 64 x 96), same LDS images, same LDS-DMA schedule; what changes:
//   * a 16x16x32 MFMA contracts the WHOLE 32-k LDS row of 16 rows: lane l reads row l & 15, block l >> 4 (hi chunk
//     2 (l >> 4), lo chunk + 1, XOR acx_swz8(row) as the DMA wrote it: conflict-free for the lane groups of ds_read_b128);
//   * a k-tile is ONE step of 3 x 4 x 6 = 72 MFMAs of 16 cycles.  It runs as two halves over the weight blocks (n blocks
//     0-2, then 3-5) so that the fragment registers still rotate: the activation fragments (4 blocks x hi / lo) live for the
//     whole tile and are double-buffered (A0 / A1, the loop is unrolled by two), the weight fragments of a half are re-read
//     for the next tile as soon as their half is done.  208 of the 256 registers: 96 accumulators, 64 + 48 fragments;
//   * D = W A^T tile: lane l holds m = l & 15, n = 4 (l >> 4) .. + 3 -- again four consecutive n per lane (16-byte
//     accesses); the two lanes l, l + 16 that share an S16 block of 8 n trade halves with v_permlane16_swap.
// MI = 16-row blocks per wave: 4 (256 x 192 tiles), or 2 / 1 (128 / 64 rows) for launches whose narrower tiles all find a CU at once
// (small batches); an output element sees the same MFMAs in the same order whatever MI is
template <int EPI, int GATHER, int MI>
__global__ __launch_bounds__(512) void gemm_split16_kernel(GemmSParams p) {
    constexpr int kBM = 64 * MI, BN = 192, WN = 2, NW = 8;
    constexpr int NJ = 6;                                                 // 16-row blocks of a wave's (16 MI) x 96 tile
    static_assert(MI == 1 || MI == 2 || MI == 4, "rows per wave");
    constexpr int A_TILE = kBM * kSRowBytes, B_TILE = BN * kSRowBytes;
    constexpr int A_DMA = kBM / (8 * NW), B_DMA = BN / (8 * NW);          // 4 + 3 one-KB pieces per wave per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + 3 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    ACX_CLAIM_VGPR(255);          // CU-exclusive: 8 waves x 256 registers
    long long lid = blockIdx.x;
    {   // XCD-contiguous tile order (see gemm.hip)
        const long long nwg = gridDim.x, per = (nwg + 7) >> 3, full = nwg - (per - 1) * 8;
        const long long xcd = lid & 7, k = lid >> 3;
        lid = (xcd < full ? xcd * per : full * per + (xcd - full) * (per - 1)) + k;
    }
    const int tile_n = (int)(lid % p.tiles_n);
    const long long tile_m = lid / p.tiles_n;
    const long long m0 = tile_m * kBM;
    const int n0 = tile_n * BN;

    const int prow = lane >> 3, pchunk = lane & 7;
    const char* a_src[A_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int row = A_DMA * 8 * wave + 8 * i + prow;
        const int chunk = pchunk ^ acx_swz8(row);
        long long m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        if (GATHER) {
            const int wo = (int)(m % p.Wo);
            const long long t = m / p.Wo;
            const int ho = (int)(t % p.Ho);
            const long long b = t / p.Ho;
            a_src[i] = p.A + (((b * p.H + 2 * ho) * p.W + 2 * wo) * p.C) * 4 + 16 * chunk;
        } else {
            a_src[i] = p.A + m * p.K * 4 + 16 * chunk;
        }
    }
    const char* b_src[B_DMA];
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int row = B_DMA * 8 * wave + 8 * i + prow;
        const int chunk = pchunk ^ acx_swz8(row);
        b_src[i] = p.Wt + (long long)(n0 + row) * p.K * 4 + 16 * chunk;
    }
    char* a_dst = As + A_DMA * 8 * wave * kSRowBytes;
    char* b_dst = Bs + B_DMA * 8 * wave * kSRowBytes;
    auto a_koff = [&](int k0) -> long long {        // byte offset of k-tile k0 inside an A row
        if (GATHER) {
            const int qd = k0 / p.C;
            return ((long long)((qd >> 1) * p.W + (qd & 1)) * p.C + (k0 - qd * p.C)) * 4;
        }
        return (long long)k0 * 4;
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int sw = acx_swz8(l15);
    const int foff_hi = l15 * kSRowBytes + (((2 * g4) ^ sw) << 4);
    const int foff_lo = l15 * kSRowBytes + (((2 * g4 + 1) ^ sw) << 4);
    const int a_frag_off = wm * (16 * MI) * kSRowBytes;
    const int b_frag_off = wn * 96 * kSRowBytes;
#define ACX_H8(x) __builtin_bit_cast(h8, x)
#define ACX_RD_A(F, abase)                                                                             \
    {   _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                               \
            F[i][0] = *reinterpret_cast<const f32x4*>((abase) + i * 16 * kSRowBytes + foff_hi);        \
            F[i][1] = *reinterpret_cast<const f32x4*>((abase) + i * 16 * kSRowBytes + foff_lo); } }
#define ACX_RD_B(F, bbase, j0)                                                                         \
    {   _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                \
            F[j][0] = *reinterpret_cast<const f32x4*>((bbase) + ((j0) + j) * 16 * kSRowBytes + foff_hi); \
            F[j][1] = *reinterpret_cast<const f32x4*>((bbase) + ((j0) + j) * 16 * kSRowBytes + foff_lo); } }
    // term 0: W lo x A hi, 1: W hi x A lo, 2: W hi x A hi; term-major: MFMAs on one accumulator are 12 instructions apart
#define ACX_MFMA16(AF, BF, j0, term, i, j)                                                             \
    acc[i][(j0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ACX_H8(BF[j][(term) == 0 ? 1 : 0]),     \
                                                              ACX_H8(AF[i][(term) == 1 ? 1 : 0]), acc[i][(j0) + j], 0, 0, 0);
#define ACX_MFMA_HALF(AF, BF, j0)                                                                      \
    {   _Pragma("unroll") for (int term = 0; term < 3; ++term)                                         \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) { ACX_MFMA16(AF, BF, j0, term, i, j) } }
    // Behind a 16x16x32 MFMA two single-issue instructions are free (tools/lab/coissue2.hip): the fragment reads and the LDS-DMA
    // pieces are spread over the MFMA gaps, one read per gap, instead of standing in clumps between the halves.
    // half 0 of tile t: behind MFMA mc < 6 the read of weight fragment 3 + mc / 2 (hi / lo) of tile t itself -- the registers
    // half 1 will use (they were busy until the previous tile's half 1 ended)
#define ACX_MFMA_HALF0_RD(AF, BF, BFH, bcur)                                                           \
    {   _Pragma("unroll") for (int term = 0; term < 3; ++term)                                         \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                \
            const int mc = (term * MI + i) * 3 + j;                                                    \
            ACX_MFMA16(AF, BF, 0, term, i, j)                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            if (mc < 6) BFH[mc >> 1][mc & 1] = *reinterpret_cast<const f32x4*>((bcur) + (3 + (mc >> 1)) * 16 * kSRowBytes + ((mc & 1) ? foff_lo : foff_hi)); \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        } }
    // half 1 of tile t: behind MFMA mc < 14 one fragment read of tile t + 1 (8 activation, 6 weight: the registers half 0 of
    // the next tile will use), behind every second MFMA one LDS-DMA piece: first the B pieces of tile t+2, then -- behind one
    // fence -- the A pieces of tile t+3 (the counted wait at the next barrier relies on this order)
#define ACX_MFMA_HALF1_DMA_RD(AF, BF, ANX, BLX, abn_, bbn_, koffA, aslot, k0B, bslot)                  \
    {   _Pragma("unroll") for (int term = 0; term < 3; ++term)                                         \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                \
            const int mc = (term * MI + i) * 3 + j;                                                    \
            ACX_MFMA16(AF, BF, 3, term, i, j)                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            if ((mc & 1) == 0) {                                                                       \
                const int pc = mc >> 1;                                                                \
                if (pc < B_DMA) lds_dma16_s(b_src[pc] + (k0B), b_dst + (bslot) * B_TILE + pc * 8 * kSRowBytes); \
                else if (pc < A_DMA + B_DMA)                                                           \
                    lds_dma16_s(a_src[pc - B_DMA] + (koffA), a_dst + (aslot) * A_TILE + (pc - B_DMA) * 8 * kSRowBytes); \
            }                                                                                          \
            if (mc < 2 * MI) ANX[mc >> 1][mc & 1] = *reinterpret_cast<const f32x4*>((abn_) + (mc >> 1) * 16 * kSRowBytes + ((mc & 1) ? foff_lo : foff_hi)); \
            else if (mc < 2 * MI + 6) BLX[(mc - 2 * MI) >> 1][mc & 1] = *reinterpret_cast<const f32x4*>((bbn_) + ((mc - 2 * MI) >> 1) * 16 * kSRowBytes + ((mc & 1) ? foff_lo : foff_hi)); \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        } }
#define ACX_DMA_A(koffA, aslot)                                                                        \
    {   _Pragma("unroll") for (int i = 0; i < A_DMA; ++i)                                              \
            lds_dma16_s(a_src[i] + (koffA), a_dst + (aslot) * A_TILE + i * 8 * kSRowBytes); }
#define ACX_DMA_B(k0B, bslot)                                                                          \
    {   _Pragma("unroll") for (int i = 0; i < B_DMA; ++i)                                              \
            lds_dma16_s(b_src[i] + (k0B), b_dst + (bslot) * B_TILE + i * 8 * kSRowBytes); }
#define ACX_TOUCH_A(F) { _Pragma("unroll") for (int i = 0; i < MI; ++i) { asm volatile("" :: "v"(F[i][0])); asm volatile("" :: "v"(F[i][1])); } }
#define ACX_TOUCH_B(F) { _Pragma("unroll") for (int j = 0; j < 3; ++j) { asm volatile("" :: "v"(F[j][0])); asm volatile("" :: "v"(F[j][1])); } }

    const int nk = p.K / kSBK;                                             // even, >= 2 (checked by the launcher)
    auto ktile = [&](int t) { return (t < nk ? t : nk - 1) * kSBK; };      // past the end: re-request the last tile
    ACX_DMA_A(a_koff(0), 0)
    ACX_DMA_B(0LL, 0)
    __syncthreads();
    ACX_DMA_B((long long)ktile(1) * 4, 1)
    ACX_DMA_A(a_koff(ktile(1)), 1)
    ACX_DMA_A(a_koff(ktile(2)), 2)
    f32x4 A0[MI][2], A1[MI][2], BL[3][2], BH[3][2];
    {
        const char* ab = As + a_frag_off;
        const char* bb = Bs + b_frag_off;
        ACX_RD_A(A0, ab)
        ACX_RD_B(BL, bb, 0)
    }
    // steady state, tile t (A slot t % 3, B slot t & 1), activation fragments AC (this tile) / AN (the next one):
    //   MFMA half 0 | counted wait + barrier (tile t+1 landed, tile t read by every wave) | rd A(t+1), B 0-2 (t+1) |
    //   MFMA half 1 threaded with DMA B(t+2) -> B slot of t, then DMA A(t+3) -> A slot of t | rd B 3-5 (t+1)
    int a_cur = 0;
#define ACX_BODY(kt_, AC, AN)                                                                          \
    {                                                                                                  \
        const int a_nxt = a_cur == 2 ? 0 : a_cur + 1;                                                  \
        const char* abn = As + a_nxt * A_TILE + a_frag_off;                                            \
        const char* bbc = Bs + ((kt_) & 1) * B_TILE + b_frag_off;                                      \
        const char* bbn = Bs + (((kt_) + 1) & 1) * B_TILE + b_frag_off;                                \
        const long long kb = (long long)ktile((kt_) + 2) * 4;                                          \
        const long long ka = a_koff(ktile((kt_) + 3));                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        ACX_MFMA_HALF0_RD(AC, BL, BH, bbc)                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        ACX_TOUCH_B(BH)                     /* tile t is fully in registers before its slots are given away */ \
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_DMA) : "memory");                                  \
        __builtin_amdgcn_s_barrier();                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        ACX_MFMA_HALF1_DMA_RD(AC, BH, AN, BL, abn, bbn, ka, a_cur, kb, (kt_) & 1)                      \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        ACX_TOUCH_A(AN)                                                                                \
        ACX_TOUCH_B(BL)                                                                                \
        a_cur = a_nxt;                                                                                 \
    }
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        ACX_BODY(kt, A0, A1)
        ACX_BODY(kt + 1, A1, A0)
    }
    ACX_BODY(kt, A0, A1)                  // kt = nk - 2: the last tile's fragments land in A1 / BL
    {
        const char* bbc = Bs + ((nk - 1) & 1) * B_TILE + b_frag_off;
        ACX_MFMA_HALF0_RD(A1, BL, BH, bbc)
    }
    ACX_MFMA_HALF(A1, BH, 3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the repeated requests of the last iterations: nothing may still be writing the LDS when the workgroup ends
#undef ACX_BODY
#undef ACX_RD_A
#undef ACX_RD_B
#undef ACX_MFMA16
#undef ACX_MFMA_HALF
#undef ACX_MFMA_HALF0_RD
#undef ACX_MFMA_HALF1_DMA_RD
#undef ACX_DMA_A
#undef ACX_DMA_B
#undef ACX_TOUCH_A
#undef ACX_TOUCH_B
#undef ACX_H8

    // ---- epilogues: lane (l15, g4) owns row m = m0 + 64 wm + 16 i + l15, columns nb + 4 g4 .. + 3 of block (i, j) ------
    // Round 4: the bias vectors are loaded ONCE, ahead of the loops, and a tile that lies inside M stores without a mask.  Before,
    // every (i, j) step loaded its bias again and stored behind `if (ok)`: a load, a wait and a store per step, each in a basic
    // block of its own, where hipcc can only wait for ALL outstanding vector-memory operations -- the previous store included:
    // 24 dependent L2 round trips per thread and tile with nothing else on the CU to hide them (tools/lab/isa_skeleton.py).
    const float sinv = p.sinv;
    f32x4 bj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bj[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 96 + j * 16 + 4 * g4);
    auto epilogue = [&](auto masked) __attribute__((always_inline)) {
    constexpr bool kMasked = decltype(masked)::value;
    if (EPI == 1) {
        const GeluK3 gk = gelu_k3(sinv, p.hscale);      // GELU of v = a * sinv, result x p.hscale (split_math.h, third form)
        const float binv = 1.0f / sinv;     // a power of two
        // One S16 block (8 n) = [hi x8][lo x8] is shared by the lanes (g4, g4 ^ 1) of a pixel row: after a permlane16
        // swap the even lane holds all 8 hi halves and the odd lane all 8 lo halves -> one 16-B store each.
        char* outb = reinterpret_cast<char*>(p.out);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const long long m = m0 + wm * (16 * MI) + i * 16 + l15;
            const bool ok = !kMasked || m < p.M;
            char* orow = outb + (ok ? m : 0) * p.N * 4;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nb = n0 + wn * 96 + j * 16;
                const f32x4 b4 = bj[j];
                unsigned xh[2], xl[2];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {      // acc holds v / sinv - bias / sinv: add the pre-scaled bias first
                    const float ax = acc[i][j][2 * e2] + b4[2 * e2] * binv;
                    const float ay = acc[i][j][2 * e2 + 1] + b4[2 * e2 + 1] * binv;
                    gelu3_pair(gk, ax, ay, xh[e2], xl[e2]);
                }
                // even rows of 16 lanes: (own hi, partner hi); odd rows: (partner lo, own lo)
                auto r0 = __builtin_amdgcn_permlane16_swap(xh[0], xl[0], false, false);
                auto r1 = __builtin_amdgcn_permlane16_swap(xh[1], xl[1], false, false);
                uint4 o;
                o.x = r0[0]; o.y = r1[0]; o.z = r0[1]; o.w = r1[1];
                if (ok) *reinterpret_cast<uint4*>(orow + (long long)(nb + 8 * (g4 >> 1)) * 4 + 16 * (g4 & 1)) = o;
            }
        }
    } else {
        float* outf = reinterpret_cast<float*>(p.out);
        // EPI 2: the residual values of row block i + 1 are requested before row block i is stored (two register sets)
        f32x4 rv[2][NJ];
        auto load_resid = [&](const int i) __attribute__((always_inline)) {
            const long long m = m0 + wm * (16 * MI) + i * 16 + l15;
            const long long row = ((!kMasked || m < p.M) ? m : 0) * p.N;
#pragma unroll
            for (int j = 0; j < NJ; ++j) rv[i & 1][j] = *reinterpret_cast<const f32x4*>(p.resid + row + n0 + wn * 96 + j * 16 + 4 * g4);
        };
        if (EPI == 2) load_resid(0);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const long long m = m0 + wm * (16 * MI) + i * 16 + l15;
            const bool ok = !kMasked || m < p.M;
            const long long row = (ok ? m : 0) * p.N;
            if (EPI == 2 && i + 1 < MI) load_resid(i + 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nb = n0 + wn * 96 + j * 16;
                const f32x4 b4 = bj[j];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(acc[i][j][e], sinv, b4[e]);
                    if (EPI == 2) v[e] += rv[i & 1][j][e];
                }
                if (ok) *reinterpret_cast<f32x4*>(outf + row + nb + 4 * g4) = v;
            }
        }
    }
    };
    if (m0 + 64 * MI <= p.M) epilogue(std::false_type{});
    else epilogue(std::true_type{});
}

template <int EPI, int GATHER, int MI>
static int launch_s16_cfg(const GemmSParams& p0, hipStream_t s) {
    GemmSParams p = p0;
    p.tiles_n = p.N / 192;
    const long long tiles_m = (p.M + 64 * MI - 1) / (64 * MI);
    const long long blocks = tiles_m * p.tiles_n;
    if (blocks > 0x7fffffffLL) ACX_FAIL(ACX_ERR_SHAPE, "gemm_split: grid too large");
    constexpr size_t lds = kCuLdsBytes;          // all of it: CU-exclusive
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &gemm_split16_kernel<EPI, GATHER, MI>, lds));
    launch_kernel(&gemm_split16_kernel<EPI, GATHER, MI>, dim3((unsigned)blocks), dim3(512), lds, s, p);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}
// the narrowest row tile (64 / 128 / 256 rows) whose workgroups -- those of every sub-batch in flight -- all find a CU at once:
// a small launch spreads over more CUs; beyond one round the 256-row tile is the efficient one.  ACX_GEMM_MI forces one (tests).
template <int EPI, int GATHER>
static int launch_s16_any(const GemmSParams& p, int ways, hipStream_t s) {
    int cus = 0;
    ACX_TRY(cu_count_of_current_device(&cus));
    const long long tn = p.N / 192;
    int mi = 4;
    if ((p.M + 63) / 64 * tn * ways <= cus) mi = 1;
    else if ((p.M + 127) / 128 * tn * ways <= cus) mi = 2;
    if (const int f = tuning().gemm_mi.load(std::memory_order_relaxed)) mi = f;
    if (mi == 1) return launch_s16_cfg<EPI, GATHER, 1>(p, s);
    if (mi == 2) return launch_s16_cfg<EPI, GATHER, 2>(p, s);
    return launch_s16_cfg<EPI, GATHER, 4>(p, s);
}

template <int kBM, int BN, int WM, int WN, int EPI, int GATHER>
static int launch_s_cfg(const GemmSParams& p0, hipStream_t s) {
    GemmSParams p = p0;
    p.tiles_n = p.N / BN;
    const long long tiles_m = (p.M + kBM - 1) / kBM;
    const long long blocks = tiles_m * p.tiles_n;
    if (blocks > 0x7fffffffLL) ACX_FAIL(ACX_ERR_SHAPE, "gemm_split: grid too large");
    static_assert((size_t)(3 * kBM + 2 * BN) * kSRowBytes <= kCuLdsBytes, "tile does not fit the LDS");
    constexpr size_t lds = kCuLdsBytes;          // all of it: CU-exclusive (see the header comment)
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &gemm_split_kernel<kBM, BN, WM, WN, EPI, GATHER>, lds));
    launch_kernel(&gemm_split_kernel<kBM, BN, WM, WN, EPI, GATHER>, dim3((unsigned)blocks), dim3(64 * WM * WN), lds, s, p);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

template <int EPI, int GATHER>
static int launch_s_bn(const GemmSParams& p, int ways, hipStream_t s) {
    // every N of the model (192, 384, 768, 1536, 3072) is a multiple of 192: 256 x 192 tiles -- 36 MFMAs per barrier and
    // the fewest operand bytes per flop through the LDS-DMA path; 256 x 128 for other multiples of 128
    if (p.N % 192 == 0) {
        const bool old_shape = tuning().gemm_32x32.load(std::memory_order_relaxed) == 1;   // A/B switch
        if ((p.K / kSBK) % 2 == 0 && !old_shape) return launch_s16_any<EPI, GATHER>(p, ways, s);
        return launch_s_cfg<256, 192, 4, 2, EPI, GATHER>(p, s);
    }
    if (p.N % 128 == 0) return launch_s_cfg<256, 128, 4, 2, EPI, GATHER>(p, s);
    ACX_FAIL(ACX_ERR_SHAPE, "gemm_split: N=%d is not a multiple of 192 or 128", p.N);
}

int launch_gemm_split(acx_ctx* c, const GemmSplitArgs& a, hipStream_t s) {
    if (a.K % kSBK != 0 || a.K < 2 * kSBK) ACX_FAIL(ACX_ERR_SHAPE, "gemm_split: K=%d is not a multiple of %d >= %d", a.K, kSBK, 2 * kSBK);
    if (a.M <= 0) return ACX_OK;
    GemmSParams p;
    p.A = reinterpret_cast<const char*>(a.A); p.Wt = reinterpret_cast<const char*>(a.Wt); p.bias = a.bias;
    p.out = a.out; p.resid = a.resid; p.M = a.M; p.N = a.N; p.K = a.K; p.sinv = a.sinv; p.hscale = a.hscale;
    p.H = a.H; p.W = a.W; p.C = a.C; p.Ho = a.Ho; p.Wo = a.Wo; p.tiles_n = 0;
    ProfScope ps(c, a.cls, s);
    const int ways = inflight_ways();
    if (a.gather) {
        if (a.epi != EPI_BIAS || a.C % kSBK != 0) ACX_FAIL(ACX_ERR_ARG, "gemm_split: bad gather configuration");
        return launch_s_bn<0, 1>(p, ways, s);
    }
    if (a.epi == EPI_GELU) return launch_s_bn<1, 0>(p, ways, s);
    if (a.epi == EPI_RESID) return launch_s_bn<2, 0>(p, ways, s);
    if (a.epi == EPI_BIAS) return launch_s_bn<0, 0>(p, ways, s);
    ACX_FAIL(ACX_ERR_ARG, "gemm_split: unknown epilogue %d", a.epi);
}

}  // namespace acx
