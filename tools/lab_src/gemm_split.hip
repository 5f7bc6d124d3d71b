// K4s/K5s -- the dense contractions with fp32 operands carried as TWO fp16 halves (ACX_PREC_F32_SPLIT).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the fp16 matrix rate on gfx950 (157 vs 2500 TFLOP/s).  An fp32 value v
// is v = hi + lo + e with hi = fp16(v), lo = fp16(v - hi), |e| <= 2^-24 |v| (two round-to-nearest 11-bit pieces
// cover 24 significant bits; gfx950 MFMA honours fp16 subnormals -- tools/mfma_denorm_probe.hip -- and a
// power-of-two pre-scale keeps the pieces far from the bottom of the fp16 range).  A product of two such values is
//   a b = ah bh + ah bl + al bh + (al bl ~ 2^-24 a b, dropped),
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
// Tiling / staging as gemm.hip: 128 x BN x 32 per workgroup, both operands by LDS-DMA with the XOR swizzle on the
// source address, fragments double-buffered in registers.  A k-tile is only 2 x TM*TN*3 MFMAs of 32 cycles, so
// the pipeline is one k-tile deeper than the fp32 kernel's: tile t+2 is in flight while tile t is multiplied.
#include "acx_internal.h"
#include "split_math.h"

namespace acx {


constexpr int kSRowBytes = 128;     // 32 k per LDS row
constexpr int kSBK = 32;

#ifdef ACX_SLAB_CLOCK       // diagnostic build (tools/split_lab.hip): in-kernel shader clock = d s_memtime / d s_memrealtime
__device__ unsigned long long acx_gs_clock[4];
#endif

struct GemmSParams {
    const char* A; const char* Wt; const float* bias; void* out; const float* resid;
    long long M; int N; int K;
    float sinv;                // 1 / (scale of A * scale of Wt)
    float hscale;              // EPI 1: scale of the S16 result (power of two)
    int H, W, C, Ho, Wo;       // gather mode: A is (B,H,W,C) S16 rows; row m = (b,ho,wo), k = (dy*2+dx)*C + c
    int tiles_n;
};

__device__ __forceinline__ void lds_dma16_s(const char* gsrc, char* lds_wave_base) {
#ifdef ACX_SLAB_NO_DMA      // diagnostic: no operand traffic at all (LDS holds garbage)
    return;
#endif
#ifdef ACX_DBG_SYNC_STAGE   // diagnostic: register staging (global_load -> ds_write_b128) instead of LDS-DMA, same layout
    {
        const f32x4 v = *reinterpret_cast<const f32x4*>(gsrc);
        *reinterpret_cast<f32x4*>(lds_wave_base + 16 * (threadIdx.x & 63)) = v;
        return;
    }
#endif
#ifdef ACX_DBG_PLAIN_LOADS  // diagnostic: the same bytes by ordinary 16-B loads into registers (LDS keeps garbage)
    {
        const f32x4 v = *reinterpret_cast<const f32x4*>(gsrc + 16 * (threadIdx.x & 63) * 0);
        asm volatile("" :: "v"(v));
        return;
    }
#endif
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
    char* Bs = smem + 2 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
#ifdef ACX_DBG_EXCL        // diagnostic: every wave claims the SIMD's whole register file (512 = v255 + a255)
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, v255" ::: "v255", "a255");
#endif
#ifdef ACX_SLAB_CLOCK
    unsigned long long ck0 = 0, rt0 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck0), "=s"(rt0) :: "memory");
#endif
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
        const int chunk = pchunk ^ ((row >> 1) & 7);
        long long m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        if (GATHER) {
            const int wo = (int)(m % p.Wo);
            const long long t = m / p.Wo;
            const int ho = (int)(t % p.Ho);
            const long long b = t / p.Ho;
            a_src[i] = p.A + (((b * p.H + 2 * ho) * p.W + 2 * wo) * p.C) * 4 + 16 * chunk;
        } else {
#ifdef ACX_SLAB_A_RESIDENT     // diagnostic: the A stream comes from L2 (256 distinct rows), not from HBM
            a_src[i] = p.A + (m & 255) * p.K * 4 + 16 * chunk;
#else
            a_src[i] = p.A + m * p.K * 4 + 16 * chunk;
#endif
        }
    }
    const char* b_src[B_DMA];
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int row = B_DMA * 8 * wave + 8 * i + prow;
        const int chunk = pchunk ^ ((row >> 1) & 7);
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

    const int sw = (l31 >> 1) & 7;
    int foff_hi[2], foff_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        foff_hi[s] = l31 * kSRowBytes + (((4 * s + 2 * hh) ^ sw) << 4);
        foff_lo[s] = l31 * kSRowBytes + (((4 * s + 2 * hh + 1) ^ sw) << 4);
    }
    const int a_frag_off = wm * TM * 32 * kSRowBytes;
    const int b_frag_off = wn * TN * 32 * kSRowBytes;
#ifdef ACX_DBG_NO_LDSREAD
#define ACX_READ_FRAGS(F, abase, bbase, s) { _Pragma("unroll") for (int i = 0; i < TM; ++i) { asm volatile("" : "+v"(F##ah[i])); asm volatile("" : "+v"(F##al[i])); } _Pragma("unroll") for (int j = 0; j < TN; ++j) { asm volatile("" : "+v"(F##bh[j])); asm volatile("" : "+v"(F##bl[j])); } }
#else
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
#endif
#define ACX_H8(x) __builtin_bit_cast(h8, x)
#ifdef ACX_SLAB_NO_MFMA
#define ACX_MFMA1(term, i, j, F) asm volatile("" :: "v"(F##ah[i]), "v"(F##bl[j]), "v"(F##al[i]), "v"(F##bh[j]));
#elif defined(ACX_DBG_BF16_MFMA)     /* diagnostic: same kernel, bf16 opcode (numerically meaningless) */
typedef __bf16 dbg_b8 __attribute__((ext_vector_type(8)));
#define ACX_MFMA1(term, i, j, F)                                                                       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dbg_b8, (term) == 0 ? F##bl[j] : F##bh[j]),  \
                                                            __builtin_bit_cast(dbg_b8, (term) == 1 ? F##al[i] : F##ah[i]), acc[i][j], 0, 0, 0);
#elif defined(ACX_DBG_TWO_TERM)
#define ACX_MFMA1(term, i, j, F)                                                                       \
    if ((term) >= 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(F##bh[j]), ACX_H8((term) == 1 ? F##al[i] : F##ah[i]), acc[i][j], 0, 0, 0); \
    else asm volatile("" :: "v"(F##bl[j]));
#elif defined(ACX_DBG_NOP_TERM)
#define ACX_MFMA1(term, i, j, F)                                                                       \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8((term) == 0 ? F##bl[j] : F##bh[j]),     \
                                                       ACX_H8((term) == 1 ? F##al[i] : F##ah[i]), acc[i][j], 0, 0, 0); \
    __builtin_amdgcn_s_nop(7);
#elif defined(ACX_DBG_SHAPE16)      /* diagnostic: the same loop with 16x16x32 MFMAs, two per 32x32x16 (same flops, same
                                       operands and LDS reads; numerically meaningless) -- what would the other shape buy? */
typedef float dbg_f4 __attribute__((ext_vector_type(4)));
#define ACX_MFMA1(term, i, j, F)                                                                       \
    {                                                                                                  \
        dbg_f4 lo_ = __builtin_shufflevector(acc[i][j], acc[i][j], 0, 1, 2, 3);                        \
        dbg_f4 hi_ = __builtin_shufflevector(acc[i][j], acc[i][j], 4, 5, 6, 7);                        \
        lo_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ACX_H8((term) == 0 ? F##bl[j] : F##bh[j]), ACX_H8((term) == 1 ? F##al[i] : F##ah[i]), lo_, 0, 0, 0); \
        hi_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ACX_H8((term) == 1 ? F##al[i] : F##ah[i]), ACX_H8((term) == 0 ? F##bl[j] : F##bh[j]), hi_, 0, 0, 0); \
        acc[i][j][0] = lo_[0]; acc[i][j][1] = lo_[1]; acc[i][j][2] = lo_[2]; acc[i][j][3] = lo_[3];    \
        acc[i][j][4] = hi_[0]; acc[i][j][5] = hi_[1]; acc[i][j][6] = hi_[2]; acc[i][j][7] = hi_[3];    \
    }
#elif defined(ACX_DBG_ONE_TERM)
#define ACX_MFMA1(term, i, j, F)                                                                       \
    if ((term) == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(F##bh[j]), ACX_H8(F##ah[i]), acc[i][j], 0, 0, 0); \
    else asm volatile("" :: "v"(F##bl[j]), "v"(F##al[i]));
#else
#define ACX_MFMA1(term, i, j, F)     /* term 0: lo x hi, 1: hi x lo, 2: hi x hi */                      \
    if (SWAP) {                                                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8((term) == 0 ? F##bl[j] : F##bh[j]),  \
                                                           ACX_H8((term) == 1 ? F##al[i] : F##ah[i]), acc[i][j], 0, 0, 0); \
    } else {                                                                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8((term) == 0 ? F##al[i] : F##ah[i]),  \
                                                           ACX_H8((term) == 1 ? F##bl[j] : F##bh[j]), acc[i][j], 0, 0, 0); \
    }
#endif
    // term-major order: MFMAs on the same accumulator are TM*TN instructions apart
#ifdef ACX_SLAB_SETPRIO
#define ACX_PRIO_HI __builtin_amdgcn_s_setprio(3);
#define ACX_PRIO_LO __builtin_amdgcn_s_setprio(0);
#else
#define ACX_PRIO_HI
#define ACX_PRIO_LO
#endif
#define ACX_MFMA_STEP(F)                                                                               \
    {                                                                                                  \
        ACX_PRIO_HI                                                                                    \
        _Pragma("unroll") for (int term = 0; term < 3; ++term)                                         \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) { ACX_MFMA1(term, i, j, F) }                    \
        ACX_PRIO_LO                                                                                    \
    }
    // the same with the LDS-DMA pieces of a later tile threaded in, one piece in front of each MFMA
#define ACX_MFMA_STEP_DMA(F, koffA, k0B, buf)                                                          \
    {                                                                                                  \
        static_assert(A_DMA + B_DMA <= 3 * TM * TN, "one DMA piece per MFMA");                         \
        _Pragma("unroll") for (int term = 0; term < 3; ++term)                                         \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                               \
            const int pc = (term * TM + i) * TN + j;                                                   \
            if (pc < A_DMA) lds_dma16_s(a_src[pc] + (koffA), a_dst + (buf) * A_TILE + pc * 8 * kSRowBytes); \
            else if (pc < A_DMA + B_DMA)                                                               \
                lds_dma16_s(b_src[pc - A_DMA] + (k0B), b_dst + (buf) * B_TILE + (pc - A_DMA) * 8 * kSRowBytes); \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            ACX_MFMA1(term, i, j, F)                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
    }
#define ACX_DMA_TILE(koffA, k0B, buf)                                                                  \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < A_DMA; ++i)                                              \
            lds_dma16_s(a_src[i] + (koffA), a_dst + (buf) * A_TILE + i * 8 * kSRowBytes);              \
        _Pragma("unroll") for (int i = 0; i < B_DMA; ++i)                                              \
            lds_dma16_s(b_src[i] + (k0B), b_dst + (buf) * B_TILE + i * 8 * kSRowBytes);                \
    }
#define ACX_TOUCH(F)      /* see gemm.hip: keeps hipcc's lgkmcnt(0) off freshly issued reads */        \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) { asm volatile("" :: "v"(F##ah[i])); asm volatile("" :: "v"(F##al[i])); } \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) { asm volatile("" :: "v"(F##bh[j])); asm volatile("" :: "v"(F##bl[j])); } \
    }

#ifdef ACX_SLAB_NO_BARRIER     // diagnostic: k-loop without its barrier (wrong results)
#define ACX_LOOP_BARRIER
#else
#define ACX_LOOP_BARRIER __syncthreads();
#endif
    const int nk = p.K / kSBK;
    // prologue: tiles 0 and 1 in flight, fragments of tile 0 in registers
    ACX_DMA_TILE(a_koff(0), 0LL, 0)
    __syncthreads();
    ACX_DMA_TILE(a_koff(kSBK), (long long)kSBK * 4, 1)         // nk >= 2 (checked by the launcher)
    f32x4 F0ah[TM], F0al[TM], F0bh[TN], F0bl[TN], F1ah[TM], F1al[TM], F1bh[TN], F1bl[TN];
    {
        const char* ab = As + a_frag_off;
        const char* bb = Bs + b_frag_off;
        ACX_READ_FRAGS(F0, ab, bb, 0)
        ACX_READ_FRAGS(F1, ab, bb, 1)
    }
    // steady state, tile t:  MFMA s0 | barrier (tile t+1 landed, tile t fully read) | rd s0(t+1) |
    //                        MFMA s1 threaded with DMA(t+2 -> buffer of t) | rd s1(t+1)
    // (no conditional inside the loop: a branch around the DMA variant makes hipcc copy all accumulators twice
    //  per iteration)
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
        const char* abn = As + ((kt + 1) & 1) * A_TILE + a_frag_off;
        const char* bbn = Bs + ((kt + 1) & 1) * B_TILE + b_frag_off;
        const int k2 = (kt + 2) * kSBK;
        const long long ka = a_koff(k2);
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_STEP(F0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(F1)
        ACX_LOOP_BARRIER
        ACX_READ_FRAGS(F0, abn, bbn, 0)
        __builtin_amdgcn_sched_barrier(0);
#ifdef ACX_GS_ONE_IN_FLIGHT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifdef ACX_GS_BURST
        ACX_DMA_TILE(ka, (long long)k2 * 4, kt & 1)
        ACX_MFMA_STEP(F1)
#else
        ACX_MFMA_STEP_DMA(F1, ka, (long long)k2 * 4, kt & 1)
#endif
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(F0)
        ACX_READ_FRAGS(F1, abn, bbn, 1)
    }
    {   // tile nk-2: nothing left to fetch
        const char* abn = As + ((kt + 1) & 1) * A_TILE + a_frag_off;
        const char* bbn = Bs + ((kt + 1) & 1) * B_TILE + b_frag_off;
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_STEP(F0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(F1)
        ACX_LOOP_BARRIER
        ACX_READ_FRAGS(F0, abn, bbn, 0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_STEP(F1)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(F0)
        ACX_READ_FRAGS(F1, abn, bbn, 1)
    }
    ACX_MFMA_STEP(F0)
    ACX_MFMA_STEP(F1)
#undef ACX_READ_FRAGS
#undef ACX_MFMA1
#undef ACX_MFMA_STEP
#undef ACX_MFMA_STEP_DMA
#undef ACX_DMA_TILE
#undef ACX_TOUCH
#undef ACX_H8

#ifdef ACX_SLAB_CLOCK
    {
        unsigned long long ck1 = 0, rt1 = 0;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck1), "=s"(rt1) :: "memory");
        if (tid == 0) { atomicAdd(&acx_gs_clock[0], ck1 - ck0); atomicAdd(&acx_gs_clock[1], rt1 - rt0); atomicAdd(&acx_gs_clock[2], 1ULL); }
    }
#endif
#ifdef ACX_SLAB_NO_EPI      // diagnostic (tools/split_lab.hip): main loop only
    {
        float t = 0.f;
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)
            _Pragma("unroll") for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) reinterpret_cast<float*>(p.out)[tid] = t;
        return;
    }
#endif
    const float sinv = p.sinv;
    if (EPI == 1) {
        GeluConsts gk;          // GELU of v = a * sinv, result x p.hscale (see split_math.h)
        gk.ps = 0.3275911f * 0.70710678f * sinv;
        gk.cq = 0.84932180f * sinv;       // sqrt(log2(e) / 2): exp(-v^2 / 2) = exp2(-(cq a)^2)
        gk.ca = -0.5f * sinv * p.hscale;
        gk.cb = sinv * p.hscale;
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
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + nb + 8 * q + 4 * hh);
                    unsigned xh[2], xl[2];
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {      // acc holds v / sinv - bias / sinv: add the pre-scaled bias first
                        f32x2 a2, av, t, ex, g;
                        a2.x = acc[i][j][4 * q + 2 * e2] + b4[2 * e2] * binv;
                        a2.y = acc[i][j][4 * q + 2 * e2 + 1] + b4[2 * e2 + 1] * binv;
                        gelu_piece1(a2, gk, av, t, ex);
                        gelu_piece2(a2, av, t, ex, gk, g);
                        gelu_piece3(g, xh[e2], xl[e2]);
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
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + nb + 8 * q + 4 * hh);
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

template <int kBM, int BN, int WM, int WN, int EPI, int GATHER>
static int launch_s_cfg(const GemmSParams& p0, hipStream_t s) {
    GemmSParams p = p0;
    p.tiles_n = p.N / BN;
    const long long tiles_m = (p.M + kBM - 1) / kBM;
    const long long blocks = tiles_m * p.tiles_n;
    if (blocks > 0x7fffffffLL) ACX_FAIL(ACX_ERR_SHAPE, "gemm_split: grid too large");
#if defined(ACX_DBG_EXCL)        // diagnostic: the workgroup claims the CU's whole LDS
    constexpr size_t lds = 160 * 1024;
#elif defined(ACX_DBG_LDS120)      // diagnostic: one workgroup per CU and no room for a 48-KB neighbour
    constexpr size_t lds = 120 * 1024;
#elif defined(ACX_DBG_LDS80)     // diagnostic: two workgroups fill the CU's LDS
    constexpr size_t lds = 80 * 1024;
#else
    constexpr size_t lds = (size_t)2 * (kBM + BN) * kSRowBytes;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        ACX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<kBM, BN, WM, WN, EPI, GATHER>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    gemm_split_kernel<kBM, BN, WM, WN, EPI, GATHER><<<dim3((unsigned)blocks), dim3(64 * WM * WN), lds, s>>>(p);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

template <int EPI, int GATHER>
static int launch_s_bn(const GemmSParams& p, hipStream_t s) {
    const long long tiles128 = ((p.M + 127) / 128) * (p.N % 128 == 0 ? p.N / 128 : p.N / 96);
    const bool small = tiles128 < 800;
#ifdef ACX_SPLIT_BIG
    if (p.N % 256 == 0 && p.K >= 384) return launch_s_cfg<256, 256, 2, 2, EPI, GATHER>(p, s);
    if (p.N % 128 == 0 && p.K >= 384) return launch_s_cfg<256, 128, 2, 2, EPI, GATHER>(p, s);
#endif
#ifdef ACX_SPLIT_SMALL
    if (p.N % 128 == 0) return launch_s_cfg<64, 128, 2, 2, EPI, GATHER>(p, s);
#endif
#ifdef ACX_SPLIT_8W
    if (p.N % 128 == 0 && p.K >= 192) return launch_s_cfg<256, 128, 4, 2, EPI, GATHER>(p, s);
#endif
    // pwconv1 (N = 4C >= 1536 here): 128 x 192 tiles -- 15 % fewer operand bytes per flop through the LDS-DMA path
    // and 36 instead of 24 MFMAs per barrier (tools/split_lab: s2.pw1 264 vs 277 us, s3.pw1 200 vs 211)
#ifndef ACX_SPLIT_NO_W192
    if (EPI == 1 && p.N % 192 == 0 && p.N >= 768 && !small) return launch_s_cfg<128, 192, 2, 2, EPI, GATHER>(p, s);
#endif
    if (p.N % 128 == 0) {
        if (small) return launch_s_cfg<64, 128, 2, 2, EPI, GATHER>(p, s);
        return launch_s_cfg<128, 128, 2, 2, EPI, GATHER>(p, s);
    }
    if (p.N % 96 == 0) return launch_s_cfg<128, 96, 4, 1, EPI, GATHER>(p, s);
    ACX_FAIL(ACX_ERR_SHAPE, "gemm_split: N=%d is not a multiple of 96 or 128", p.N);
}

int launch_gemm_split(acx_ctx* c, const GemmSplitArgs& a, hipStream_t s) {
    if (a.K % kSBK != 0 || a.K < 2 * kSBK) ACX_FAIL(ACX_ERR_SHAPE, "gemm_split: K=%d is not a multiple of %d >= %d", a.K, kSBK, 2 * kSBK);
    if (a.M <= 0) return ACX_OK;
    GemmSParams p;
    p.A = reinterpret_cast<const char*>(a.A); p.Wt = reinterpret_cast<const char*>(a.Wt); p.bias = a.bias;
    p.out = a.out; p.resid = a.resid; p.M = a.M; p.N = a.N; p.K = a.K; p.sinv = a.sinv; p.hscale = a.hscale;
    p.H = a.H; p.W = a.W; p.C = a.C; p.Ho = a.Ho; p.Wo = a.Wo; p.tiles_n = 0;
    ProfScope ps(c, a.cls, s);
    if (a.gather) {
        if (a.epi != EPI_BIAS || a.C % kSBK != 0) ACX_FAIL(ACX_ERR_ARG, "gemm_split: bad gather configuration");
        return launch_s_bn<0, 1>(p, s);
    }
    if (a.epi == EPI_GELU) return launch_s_bn<1, 0>(p, s);
    if (a.epi == EPI_RESID) return launch_s_bn<2, 0>(p, s);
    if (a.epi == EPI_BIAS) return launch_s_bn<0, 0>(p, s);
    ACX_FAIL(ACX_ERR_ARG, "gemm_split: unknown epilogue %d", a.epi);
}

}  // namespace acx
