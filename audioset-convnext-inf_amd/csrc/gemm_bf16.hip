// bf16 precision mode (BASELINE configs[2]: "bf16 with fp32 LayerNorm"): the dense contractions on
// v_mfma_f32_32x32x16_bf16 -- bf16 operands, fp32 accumulate -- everything else (residual stream, LayerNorm
// statistics, depthwise conv, frontend, head) stays fp32.
//   A operands are bf16 in HBM: the fp32 LayerNorm pass writes its normalised rows as bf16 (ln_rows_bf16 in
//   dwconv.hip), pwconv1's epilogue writes the hidden activation as bf16.  Weights are rounded to bf16 once
//   at acx_finalize (K padded to a multiple of 64).
// Same structure as gemm.hip, as 8-wave CU-exclusive workgroups (acx_internal.h): 256 x BN tiles, 128-B LDS rows (= 64 bf16 of K), both operands by LDS-DMA with
// the XOR swizzle on the source address, fragments double-buffered in registers, one barrier per k-tile.  A
// fragment read is still one ds_read_b128: lane half h takes chunk 2g+h of its row = k 16g+8h .. +7, exactly
// the A/B lane map of the 32x32x16 instruction, so one MFMA per (tile, k-group) replaces four fp32 ones.
#include <cstdlib>
#include <type_traits>

#include "acx_internal.h"
#include "split_math.h"

namespace acx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBfRowBytes = 128;        // 64 bf16 per LDS row
constexpr int kBfBK = 64;

__device__ __forceinline__ float gelu_erf_b(float v) { return gelu3_unit(v); }     // split_math.h, third form

struct GemmBfParams {
    const __bf16* A; const __bf16* Wt; const float* bias; void* out; const float* resid;
    long long M; int N; int Kp; int lda;      // Kp: padded K (multiple of 64) = row stride of Wt; lda: row stride of A
    int H, W, Cp, Ho, Wo;                     // gather mode: A is (B,H,W,Cp) bf16, row m = (b,ho,wo), k = (dy*2+dx)*Cp + c
    int tiles_n;
};

// One 1-KB piece (8 rows x 128 B): scalar base of the tile's first row + a 32-bit offset per lane (its row's distance from that
// row + its chunk) -- the form whose issue costs one instruction instead of five (split_math.h, acx_glds16_s; round 5).
__device__ __forceinline__ void lds_dma16_b(const char* sbase, unsigned voff, char* lds_wave_base) {
    acx_glds16_s(sbase, voff, acx_lds_addr(lds_wave_base));
}

// EPI: 0 bias -> fp32, 1 bias + GELU -> bf16, 2 bias + residual -> fp32, 3 bias -> bf16 (the downsample conv feeding a stage
// whose activations live in HBM as bf16: ACX_PREC_BF16_ACT)
template <int kBM, int BN, int WM, int WN, int EPI, int GATHER>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(GemmBfParams p) {
    constexpr int TM = kBM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int NW = WM * WN;                                            // 8 waves: one CU-exclusive workgroup per CU
    constexpr int A_TILE = kBM * kBfRowBytes, B_TILE = BN * kBfRowBytes;
    constexpr int A_DMA = kBM / (8 * NW), B_DMA = BN / (8 * NW);           // 1-KB pieces per wave per tile
    static_assert(NW == 8 && A_DMA * 8 * NW == kBM && B_DMA * 8 * NW == BN, "8-wave tiles of 8-row pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + 3 * A_TILE;
    static_assert(3 * A_TILE + 2 * B_TILE <= (int)kCuLdsBytes, "the rings do not fit the LDS");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // CU-exclusive (acx_internal.h): dense 16-bit MFMA next to foreign packed-FP32 work is the same hazard as in the split
    // kernels (found by the bs = 64 two-stream bf16 test).  One 8-wave workgroup per CU, 256 registers per lane claimed.
    ACX_CLAIM_VGPR(255);
    long long lid = blockIdx.x;
    {
        const long long nwg = gridDim.x, per = (nwg + 7) >> 3, full = nwg - (per - 1) * 8;
        const long long xcd = lid & 7, k = lid >> 3;
        lid = (xcd < full ? xcd * per : full * per + (xcd - full) * (per - 1)) + k;
    }
    const int tile_n = (int)(lid % p.tiles_n);
    const long long tile_m = lid / p.tiles_n;
    const long long m0 = tile_m * kBM;
    const int n0 = tile_n * BN;

    const int prow = lane >> 3, pchunk = lane & 7;
    // element index of the first k of row m of A (rows of a tile ascend in memory in both modes: offsets from row m0 are >= 0)
    auto a_row = [&](long long m) -> long long {
        if (GATHER) {
            const int wo = (int)(m % p.Wo);
            const long long t = m / p.Wo;
            const int ho = (int)(t % p.Ho);
            const long long b = t / p.Ho;
            return ((b * p.H + 2 * ho) * p.W + 2 * wo) * p.Cp;
        }
        return m * p.lda;
    };
    const long long a_row0 = a_row(m0 < p.M ? m0 : p.M - 1);
    const char* const a_base = acx_scalar_ptr(p.A + a_row0);
    unsigned a_src[A_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int row = A_DMA * 8 * wave + 8 * i + prow;
        const int chunk = pchunk ^ ((row >> 1) & 7);
        long long m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        a_src[i] = (unsigned)((a_row(m) - a_row0 + 8 * chunk) * 2);
    }
    const char* const b_base = acx_scalar_ptr(p.Wt + (long long)n0 * p.Kp);
    unsigned b_src[B_DMA];
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int row = B_DMA * 8 * wave + 8 * i + prow;
        const int chunk = pchunk ^ ((row >> 1) & 7);
        b_src[i] = (unsigned)((row * p.Kp + 8 * chunk) * 2);
    }
    char* a_dst = As + A_DMA * 8 * wave * kBfRowBytes;
    char* b_dst = Bs + B_DMA * 8 * wave * kBfRowBytes;
    auto a_koff = [&](int k0) -> long long {        // element offset of k-tile k0 inside an A row
        if (GATHER) {
            const int qd = k0 / p.Cp;
            return (long long)((qd >> 1) * p.W + (qd & 1)) * p.Cp + (k0 - qd * p.Cp);
        }
        return (long long)k0;
    };
#define ACX_DMA_A(k0, slot)                                                                            \
    {                                                                                                  \
        const long long koff = a_koff(k0);                                                             \
        _Pragma("unroll") for (int i = 0; i < A_DMA; ++i)                                              \
            lds_dma16_b(a_base + koff * 2, a_src[i], a_dst + (slot) * A_TILE + i * 8 * kBfRowBytes);   \
    }
#define ACX_DMA_B(k0, slot)                                                                            \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < B_DMA; ++i)                                              \
            lds_dma16_b(b_base + (long long)(k0) * 2, b_src[i], b_dst + (slot) * B_TILE + i * 8 * kBfRowBytes); \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) foff[g] = l31 * kBfRowBytes + (((2 * g + hh) ^ sw) << 4);
    const int a_frag_off = wm * TM * 32 * kBfRowBytes;
    const int b_frag_off = wn * TN * 32 * kBfRowBytes;
#define ACX_READ_FRAGS(af_, bf_, abase, bbase, g)                                                      \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
            af_[i] = *reinterpret_cast<const f32x4*>((abase) + i * 32 * kBfRowBytes + foff[g]);        \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
            bf_[j] = *reinterpret_cast<const f32x4*>((bbase) + j * 32 * kBfRowBytes + foff[g]);        \
    }
#define ACX_MFMA_GROUP(af_, bf_)                                                                       \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af_[i]),    \
                                                                __builtin_bit_cast(bf16x8, bf_[j]), acc[i][j], 0, 0, 0); \
    }
    // The LDS-DMA pieces of the NEXT k-tile are threaded one by one through the MFMAs of this tile's first two groups (A pieces in
    // the first, B pieces in the second), as in gemm.hip: issued as a burst of 7 at the top of the tile they stall the wave for
    // ~700 cycles while its SIMD partner, in lockstep, stalls on its own burst (round 4; the burst form ran 0.29-0.33 of 2.5 PF).
#define ACX_MFMA_GROUP_DMA(af_, bf_, base_, src_, n_, koff_, dst_)                                     \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                               \
            if (i * TN + j < (n_)) {                                                                   \
                lds_dma16_b((base_) + (koff_) * 2, src_[i * TN + j], (dst_) + (i * TN + j) * 8 * kBfRowBytes); \
                __builtin_amdgcn_sched_barrier(0);                                                     \
            }                                                                                          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af_[i]),    \
                                                                __builtin_bit_cast(bf16x8, bf_[j]), acc[i][j], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
    }
    static_assert(A_DMA <= TM * TN && B_DMA <= TM * TN, "one piece per MFMA of a group at most");
#define ACX_TOUCH(af_, bf_)   /* see gemm.hip: hipcc only emits lgkmcnt(0) next to an LDS-DMA */      \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(af_[i]));                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bf_[j]));                \
    }

    // The k loop.  A tile is requested TWO k-tiles ahead (ring of three: in the second MFMA group of tile kt), the weights of tile
    // kt + 2 as soon as tile kt's slot is free (ring of two: behind tile kt's barrier, in its last MFMA group): the counted wait in
    // front of a barrier leaves the A pieces just requested in flight.
    // (Round 5: with both operands one tile ahead the wait in front of every barrier stood for a memory latency that 24 MFMAs per
    // wave do not cover -- a build without the wait, wrong results, ran pwconv1 / pwconv2 / the downsample convs 20-26 % faster.)
    const int nk = p.Kp / kBfBK;
    ACX_DMA_A(0, 0)
    ACX_DMA_B(0, 0)
    ACX_DMA_A((nk > 1 ? 1 : 0) * kBfBK, 1)
    ACX_DMA_B((nk > 1 ? 1 : 0) * kBfBK, 1)
    // (the compiler counts the LDS-DMA builtin's loads before a barrier by itself; the inline-asm form is invisible to it)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_DMA + B_DMA) : "memory");
    __syncthreads();
    f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
    {
        const char* ab = As + a_frag_off;
        const char* bb = Bs + b_frag_off;
        ACX_READ_FRAGS(af0, bf0, ab, bb, 0)
        ACX_READ_FRAGS(af1, bf1, ab, bb, 1)
    }
    int sa = 0;                                        // A slot of tile kt
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1;
        const char* ab = As + sa * A_TILE + a_frag_off;
        const char* bb = Bs + (kt & 1) * B_TILE + b_frag_off;
        const char* abn = As + sa1 * A_TILE + a_frag_off;
        const char* bbn = Bs + ((kt + 1) & 1) * B_TILE + b_frag_off;
        const int k2 = (kt + 2 < nk ? kt + 2 : nk - 1) * kBfBK;                      // past the end: the last tile once more, into a free slot
        const long long koff2 = a_koff(k2);
        char* adn = a_dst + sa2 * A_TILE;
        char* bdn = b_dst + (kt & 1) * B_TILE;
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP(af0, bf0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(af1, bf1)
        ACX_READ_FRAGS(af0, bf0, ab, bb, 2)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP_DMA(af1, bf1, a_base, a_src, A_DMA, koff2, adn)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(af0, bf0)
        ACX_READ_FRAGS(af1, bf1, ab, bb, 3)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP(af0, bf0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(af1, bf1)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_DMA) : "memory");      // this wave's pieces of tile kt + 1 have landed
        __syncthreads();
        ACX_READ_FRAGS(af0, bf0, abn, bbn, 0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP_DMA(af1, bf1, b_base, b_src, B_DMA, (long long)k2, bdn)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(af0, bf0)
        ACX_READ_FRAGS(af1, bf1, abn, bbn, 1)
        sa = sa1;
    }
    {
        const char* ab = As + sa * A_TILE + a_frag_off;
        const char* bb = Bs + ((nk - 1) & 1) * B_TILE + b_frag_off;
        ACX_MFMA_GROUP(af0, bf0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_READ_FRAGS(af0, bf0, ab, bb, 2)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP(af1, bf1)
        __builtin_amdgcn_sched_barrier(0);
        ACX_READ_FRAGS(af1, bf1, ab, bb, 3)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP(af0, bf0)
        ACX_MFMA_GROUP(af1, bf1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the extra request of the last tile must not land in the next workgroup's LDS
#undef ACX_DMA_A
#undef ACX_DMA_B
#undef ACX_READ_FRAGS
#undef ACX_MFMA_GROUP
#undef ACX_MFMA_GROUP_DMA
#undef ACX_TOUCH

    // ---- epilogue: D layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m) ---------------------
    // Round 4: a tile inside M runs without per-element masks, its biases are loaded once and the residual values of a 32-row
    // block are all requested before the block's first store.  Before, every element of EPI 2 was a load, a wait for ALL
    // outstanding vector-memory operations (the previous store included) and a store: 96 dependent L2 round trips per thread and
    // tile with nothing else on the CU-exclusive workgroup's CU to hide them (tools/lab/isa_skeleton.py).
    float bnj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bnj[j] = p.bias[n0 + (wn * TN + j) * 32 + l31];
    auto epilogue = [&](auto masked) __attribute__((always_inline)) {
        constexpr bool kMasked = decltype(masked)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long mb = m0 + (wm * TM + i) * 32 + 4 * hh;
            float rv[TN][16];
            if (EPI == 2) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        const long long mr = (!kMasked || mb + dr < p.M) ? mb + dr : p.M - 1;     // (clamped: the value is not used)
                        rv[j][r] = p.resid[mr * p.N + n];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + l31;
                const float bn = bnj[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (!kMasked || mb + dr < p.M) {
                        const long long off = (mb + dr) * p.N + n;
                        float v = acc[i][j][r] + bn;
                        if (EPI == 1) {
                            reinterpret_cast<__bf16*>(p.out)[off] = (__bf16)gelu_erf_b(v);
                        } else if (EPI == 3) {
                            reinterpret_cast<__bf16*>(p.out)[off] = (__bf16)v;
                        } else {
                            if (EPI == 2) v += rv[j][r];
                            reinterpret_cast<float*>(p.out)[off] = v;
                        }
                    }
                }
            }
        }
    };
    if (m0 + kBM <= p.M) epilogue(std::false_type{});
    else epilogue(std::true_type{});
}

template <int kBM, int BN, int WM, int WN, int EPI, int GATHER>
static int launch_bf_cfg(const GemmBfParams& p0, hipStream_t s) {
    GemmBfParams p = p0;
    p.tiles_n = p.N / BN;
    const long long tiles_m = (p.M + kBM - 1) / kBM;
    const long long blocks = tiles_m * p.tiles_n;
    if (blocks > 0x7fffffffLL) ACX_FAIL(ACX_ERR_SHAPE, "gemm_bf16: grid too large");
    static_assert((size_t)(3 * kBM + 2 * BN) * kBfRowBytes <= kCuLdsBytes, "rings do not fit the LDS");
    constexpr size_t lds = kCuLdsBytes;          // all of it: CU-exclusive
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &gemm_bf16_kernel<kBM, BN, WM, WN, EPI, GATHER>, lds));
    launch_kernel(&gemm_bf16_kernel<kBM, BN, WM, WN, EPI, GATHER>, dim3((unsigned)blocks), dim3(64 * WM * WN), lds, s, p);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

template <int EPI, int GATHER>
static int launch_bf_bn(const GemmBfParams& p, int ways, hipStream_t s) {
    if (p.N % 192 == 0) {                                                            // every N of the model
        // 128-row tiles when all of them -- of every sub-batch in flight -- find a CU at once (small launches spread over twice the
        // CUs; an output element sees the same MFMAs in the same order).  ACX_GEMM_MI = 1 | 2 forces them, 4 the 256-row tile.
        int cus = 0;
        ACX_TRY(cu_count_of_current_device(&cus));
        bool narrow = (p.M + 127) / 128 * (p.N / 192) * ways <= cus;
        if (const int f = tuning().gemm_mi.load(std::memory_order_relaxed)) narrow = f != 4;
        if (narrow) return launch_bf_cfg<128, 192, 4, 2, EPI, GATHER>(p, s);
        return launch_bf_cfg<256, 192, 4, 2, EPI, GATHER>(p, s);
    }
    if (p.N % 128 == 0) return launch_bf_cfg<256, 128, 4, 2, EPI, GATHER>(p, s);
    ACX_FAIL(ACX_ERR_SHAPE, "gemm_bf16: N=%d is not a multiple of 192 or 128", p.N);
}

int launch_gemm_bf16(acx_ctx* c, const GemmBf16Args& a, hipStream_t s) {
    if (a.Kp % kBfBK != 0) ACX_FAIL(ACX_ERR_SHAPE, "gemm_bf16: padded K=%d is not a multiple of %d", a.Kp, kBfBK);
    if (a.M <= 0) return ACX_OK;
    GemmBfParams p;
    p.A = reinterpret_cast<const __bf16*>(a.A); p.Wt = reinterpret_cast<const __bf16*>(a.Wt); p.bias = a.bias;
    p.out = a.out; p.resid = a.resid; p.M = a.M; p.N = a.N; p.Kp = a.Kp; p.lda = a.lda;
    p.H = a.H; p.W = a.W; p.Cp = a.Cp; p.Ho = a.Ho; p.Wo = a.Wo; p.tiles_n = 0;
    ProfScope ps(c, a.cls, s);
    const int ways = inflight_ways();
    if (a.gather) {
        if (a.epi != EPI_BIAS || a.Cp % kBfBK != 0) ACX_FAIL(ACX_ERR_ARG, "gemm_bf16: bad gather configuration");
        return a.out_bf16 ? launch_bf_bn<3, 1>(p, ways, s) : launch_bf_bn<0, 1>(p, ways, s);
    }
    if (a.out_bf16) ACX_FAIL(ACX_ERR_ARG, "gemm_bf16: bf16 output exists for the gather (downsample) form only");
    if (a.epi == EPI_GELU) return launch_bf_bn<1, 0>(p, ways, s);
    if (a.epi == EPI_RESID) return launch_bf_bn<2, 0>(p, ways, s);
    if (a.epi == EPI_BIAS) return launch_bf_bn<0, 0>(p, ways, s);
    ACX_FAIL(ACX_ERR_ARG, "gemm_bf16: unknown epilogue %d", a.epi);
}

}  // namespace acx
