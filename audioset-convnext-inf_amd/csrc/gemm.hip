// K4/K5 -- the dense contractions of the backbone on the f32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products, k-ordered fp32 accumulation):
//   pwconv1 + GELU   (Block.norm / pwconv1 / act, convnext.py:61-65, :78-80)   A = y (dwconv output)
//   pwconv2 + gamma + residual (convnext.py:66-71, :81-86)                     A = hidden
//   downsample 2x2/s2 conv as a GEMM over K = 4C (convnext.py:230-235)         A = LayerNorm(x), gathered
// out[M,N] = epi( A[M,K] . Wt[N,K]^T ).
//
// LayerNorm in front of pwconv1 is applied in the EPILOGUE: with the affine folded into Wt/bias
// (acx_finalize) LN(y).Wt^T = rstd_m * (y.Wt^T - mean_m * colsum_n), colsum_n = sum_k Wt[n][k].  That keeps
// BOTH operands plain copies, so both are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4): no
// staging VGPRs, no ds_write pass, no VALU in the load path (the register-staged version of this kernel
// spent 15 % of its time there -- profiles/r01_gemm_lab.txt).
//
// Tiling: 128 x BN x 32 per workgroup, 4 waves, each wave TM x TN tiles of 32x32, K stepped by 2 per
// MFMA.  LDS tile image: [row][8 chunks of 16 B] with 128-B rows (LDS-DMA writes wave-linear 1-KB
// pieces, so rows cannot be padded); bank conflicts are removed by an XOR swizzle applied on the per-lane
// SOURCE address and again on the fragment read: position p of row r holds global chunk p ^ ((r>>1)&7).
// A fragment read is one ds_read_b128 per 4 k-steps: lane half h takes chunk 2g+h of its row, so MFMA j
// of group g contracts k = {8g+j, 8g+4+j} (A and B use the same permutation).
#include "acx_internal.h"

namespace acx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ACX_GSTAMP(var)

constexpr int kBK = 32;
constexpr int kRowBytes = kBK * 4;        // 128-B LDS rows

// nn.GELU() default (approximate='none'): 0.5 v (1 + erf(v / sqrt 2)), with erf from Abramowitz-Stegun
// 7.1.26 (|erf error| <= 1.5e-7, so |gelu error| <= 0.75e-7 |v|):
//   erf(u) = sign(u) (1 - q),  q = (a1 t + ... + a5 t^5) exp(-u^2),  t = 1 / (1 + p |u|)
//   gelu(v) = max(v, 0) - 0.5 |v| q          (since v sign(v) = |v|)
// 14 VALU per element, two of them transcendental (v_rcp_f32, v_exp_f32).
__device__ __forceinline__ float gelu_erf(float v) {
    const float av = fabsf(v);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678f, av, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(v * v * -0.72134752f);      // exp(-v^2 / 2)
    const float q = pl * t * e;
    return fmaf(-0.5f * av, q, fmaxf(v, 0.0f));
}

struct GemmParams {
    const float* A; const float* Wt; const float* bias; float* out;
    const float* stats;       // EPI_GELU: (M,2) mean/rstd of the A rows
    const float* colsum;      // EPI_GELU: (N) sum_k Wt[n][k]
    const float* resid;       // EPI_RESID
    long long M; int N; int K;
    int H, W, C, Ho, Wo;      // gather mode (A is NHWC (B,H,W,C); row m = (b,ho,wo); k = (dy*2+dx)*C + c)
    int tiles_n;
};

__device__ __forceinline__ void lds_dma16(const float* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// GATHER: 0 plain rows, 1 2x2 patch gather (downsample)
template <int kBM, int BN, int WM, int WN, int EPI, int GATHER, int DW>
__global__ __launch_bounds__(256 + 64 * DW) void gemm_f32_kernel(GemmParams p) {
    constexpr int TM = kBM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int A_TILE = kBM * kRowBytes, B_TILE = BN * kRowBytes;     // bytes
    constexpr int A_DMA = kBM / 32, B_DMA = BN / 32;                     // 1-KB pieces per wave per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                   // [2][A_TILE]
    char* Bs = smem + 2 * A_TILE;      // [2][B_TILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so
    // logical tile id L = (bid % 8) * ceil(n/8) + bid / 8 gives every XCD a CONTIGUOUS run of tiles; with the
    // n-tile index fastest, the tiles that share an A row-panel then sit on one XCD and hit its L2 instead of
    // re-fetching the panel once per XCD (speed only: any placement is correct).
    long long lid = blockIdx.x;
    {
        const long long nwg = gridDim.x, per = (nwg + 7) >> 3, full = nwg - (per - 1) * 8;   // XCDs 0..full-1 hold `per`
        const long long xcd = lid & 7, k = lid >> 3;
        lid = (xcd < full ? xcd * per : full * per + (xcd - full) * (per - 1)) + k;
    }
    const int tile_n = (int)(lid % p.tiles_n);
    const long long tile_m = lid / p.tiles_n;
    const long long m0 = tile_m * kBM;
    const int n0 = tile_n * BN;

    const int prow = lane >> 3, pchunk = lane & 7;
    if (DW && wave == 4) {
        // ---- dedicated LDS-DMA wave: issues every 1-KB piece of both operand tiles, so the four MFMA waves
        //      never spend issue slots on loads (an LDS-DMA costs 60-180 cycles of issue, MI355X_MICROARCH.md) --
        constexpr int AP = kBM / 8, BP = BN / 8;
        const float* a_all[AP];
        const float* b_all[BP];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int row = 8 * i + prow;
            const int chunk = pchunk ^ ((row >> 1) & 7);
            long long m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            if (GATHER) {
                const int wo = (int)(m % p.Wo);
                const long long t = m / p.Wo;
                const int ho = (int)(t % p.Ho);
                const long long b = t / p.Ho;
                a_all[i] = p.A + ((b * p.H + 2 * ho) * p.W + 2 * wo) * p.C + 4 * chunk;
            } else {
                a_all[i] = p.A + m * p.K + 4 * chunk;
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int row = 8 * i + prow;
            const int chunk = pchunk ^ ((row >> 1) & 7);
            b_all[i] = p.Wt + (long long)(n0 + row) * p.K + 4 * chunk;
        }
        const int nkd = p.K / kBK;
        for (int kt = 0; kt < nkd; ++kt) {
            const int k0 = kt * kBK;
            long long koff = k0;
            if (GATHER) {
                const int qd = k0 / p.C;
                koff = (long long)((qd >> 1) * p.W + (qd & 1)) * p.C + (k0 - qd * p.C);
            }
            char* ad = As + (kt & 1) * A_TILE;
            char* bd = Bs + (kt & 1) * B_TILE;
#pragma unroll
            for (int i = 0; i < AP; ++i) lds_dma16(a_all[i] + koff, ad + i * 1024);
#pragma unroll
            for (int i = 0; i < BP; ++i) lds_dma16(b_all[i] + k0, bd + i * 1024);
            __syncthreads();      // (hipcc drains the LDS-DMA first) pairs with the MFMA waves' barrier of tile kt
        }
        return;
    }
    // ---- LDS-DMA source pointers: wave w stages rows [R*w, R*w+R) of each tile, 8 rows per piece ------
    const float* a_src[A_DMA];
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
            a_src[i] = p.A + ((b * p.H + 2 * ho) * p.W + 2 * wo) * p.C + 4 * chunk;
        } else {
            a_src[i] = p.A + m * p.K + 4 * chunk;
        }
    }
    const float* b_src[B_DMA];
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int row = B_DMA * 8 * wave + 8 * i + prow;
        const int chunk = pchunk ^ ((row >> 1) & 7);
        b_src[i] = p.Wt + (long long)(n0 + row) * p.K + 4 * chunk;
    }
    char* a_dst = As + A_DMA * 8 * wave * kRowBytes;
    char* b_dst = Bs + B_DMA * 8 * wave * kRowBytes;
#define ACX_DMA_TILE(k0, buf)                                                                          \
    {                                                                                                  \
        long long koff = (k0);                                                                         \
        if (GATHER) {                                                                                  \
            const int qd = (k0) / p.C;                                                                 \
            koff = (long long)((qd >> 1) * p.W + (qd & 1)) * p.C + ((k0) - qd * p.C);                  \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < A_DMA; ++i)                                              \
            lds_dma16(a_src[i] + koff, a_dst + (buf) * A_TILE + i * 8 * kRowBytes);                    \
        _Pragma("unroll") for (int i = 0; i < B_DMA; ++i)                                              \
            lds_dma16(b_src[i] + (k0), b_dst + (buf) * B_TILE + i * 8 * kRowBytes);                    \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (bytes) inside a tile: row l31 of the wave's 32-row tile, chunk (2g+hh)^sw
    const int sw = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) foff[g] = l31 * kRowBytes + (((2 * g + hh) ^ sw) << 4);
    const int a_frag_off = wm * TM * 32 * kRowBytes;
    const int b_frag_off = wn * TN * 32 * kRowBytes;
#define ACX_READ_FRAGS(af_, bf_, abase, bbase, g)                                                      \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
            af_[i] = *reinterpret_cast<const f32x4*>((abase) + i * 32 * kRowBytes + foff[g]);          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
            bf_[j] = *reinterpret_cast<const f32x4*>((bbase) + j * 32 * kRowBytes + foff[g]);          \
    }
#define ACX_MFMA_GROUP(af_, bf_)                                                                       \
    {                                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_[i][e], bf_[j][e], acc[i][j], 0, 0, 0); \
    }
    // MFMA group with the LDS-DMA of the NEXT tile threaded through it: one 1-KB piece in front of each of the
    // 4 k-steps (A pieces in group g0, B pieces in group g1).  Issued as a burst of 8 at the top of the tile,
    // the DMAs stall the wave for ~1k cycles while its SIMD partner -- which runs the same program and has
    // drifted into lockstep through MFMA-pipe contention -- stalls on its own burst at the same moment.
#define ACX_MFMA_GROUP_DMA(af_, bf_, src_, n_, koff_, dst_)                                            \
    {                                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                \
            if (!DW && e < (n_)) {                                                                     \
                lds_dma16(src_[e] + (koff_), (dst_) + e * 8 * kRowBytes);                              \
                __builtin_amdgcn_sched_barrier(0);                                                     \
            }                                                                                          \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                             \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_[i][e], bf_[j][e], acc[i][j], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
    }
    static_assert(A_DMA <= 4 && B_DMA <= 4, "at most 4 DMA pieces per operand per wave are threaded through a group");
    static_assert(kBK == 32, "the tile schedule below is written for 4 k-groups of 8");
    // With an LDS-DMA in flight hipcc (ROCm 7.2) no longer emits counted lgkmcnt(N) waits, only lgkmcnt(0).
    // An opaque use of the fragments the NEXT MFMA group needs, placed right after the current group (their
    // reads are 4*TM*TN MFMAs old by then), puts that wait in front of the next batch of ds_reads instead of
    // behind it, so it never covers a just-issued read.
#define ACX_TOUCH(af_, bf_)                                                                             \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(af_[i]));                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bf_[j]));                \
    }

    const int nk = p.K / kBK;
    if (!DW) ACX_DMA_TILE(0, 0);
    __syncthreads();          // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier
    // Software pipeline over k-tiles, ONE barrier per tile, placed BEFORE the last MFMA group so that
    // every wave leaves it with 4*TM*TN MFMAs already fed (barrier skew and the first LDS reads of the
    // next tile hide under them); fragments are double-buffered in registers so LDS latency never stalls
    // a wave inside a tile:
    //   DMA(t+1) | MFMA g0 | rd g2 | MFMA g1 | rd g3 | MFMA g2 | barrier | rd g0(t+1) | MFMA g3 | rd g1(t+1)
    f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
    {
        const char* ab = As + a_frag_off;
        const char* bb = Bs + b_frag_off;
        ACX_READ_FRAGS(af0, bf0, ab, bb, 0)
        ACX_READ_FRAGS(af1, bf1, ab, bb, 1)
    }
    for (int kt = 0; kt + 1 < nk; ++kt) {
        ACX_GSTAMP(g0)
        const char* ab = As + (kt & 1) * A_TILE + a_frag_off;
        const char* bb = Bs + (kt & 1) * B_TILE + b_frag_off;
        const char* abn = As + ((kt + 1) & 1) * A_TILE + a_frag_off;
        const char* bbn = Bs + ((kt + 1) & 1) * B_TILE + b_frag_off;
        const int k1 = (kt + 1) * kBK;
        long long koff1 = k1;
        if (GATHER) {
            const int qd = k1 / p.C;
            koff1 = (long long)((qd >> 1) * p.W + (qd & 1)) * p.C + (k1 - qd * p.C);
        }
        char* adn = a_dst + ((kt + 1) & 1) * A_TILE;
        char* bdn = b_dst + ((kt + 1) & 1) * B_TILE;
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP_DMA(af0, bf0, a_src, A_DMA, koff1, adn)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(af1, bf1)       // see ACX_TOUCH: drain the OLD reads before issuing new ones
        ACX_READ_FRAGS(af0, bf0, ab, bb, 2)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP_DMA(af1, bf1, b_src, B_DMA, (long long)k1, bdn)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(af0, bf0)
        ACX_READ_FRAGS(af1, bf1, ab, bb, 3)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP(af0, bf0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(af1, bf1)
        ACX_GSTAMP(g1)
        __syncthreads();
        ACX_GSTAMP(g2)
        ACX_READ_FRAGS(af0, bf0, abn, bbn, 0)
        __builtin_amdgcn_sched_barrier(0);
        ACX_MFMA_GROUP(af1, bf1)
        __builtin_amdgcn_sched_barrier(0);
        ACX_TOUCH(af0, bf0)
        ACX_READ_FRAGS(af1, bf1, abn, bbn, 1)
        ACX_GSTAMP(g3)
    }
    {
        const char* ab = As + ((nk - 1) & 1) * A_TILE + a_frag_off;
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
#undef ACX_DMA_TILE
#undef ACX_READ_FRAGS
#undef ACX_MFMA_GROUP
#undef ACX_MFMA_GROUP_DMA
#undef ACX_TOUCH

    // ---- epilogue: D tile layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m) -----
    const bool full = m0 + kBM <= p.M;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long long mb = m0 + (wm * TM + i) * 32 + 4 * hh;
        float2 st[16];
        if (EPI == EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long long m = mb + (r & 3) + 8 * (r >> 2);
                if (m >= p.M) m = p.M - 1;
                st[r] = *reinterpret_cast<const float2*>(p.stats + 2 * m);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
            const float bn = p.bias[n];
            const float cs = (EPI == EPI_GELU) ? p.colsum[n] : 0.f;
            float* op = p.out + mb * p.N + n;
            const float* rp = (EPI == EPI_RESID) ? p.resid + mb * p.N + n : nullptr;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[i][j][r];
                if (EPI == EPI_GELU) v[r] = gelu_erf(fmaf(st[r].y, fmaf(-st[r].x, cs, v[r]), bn));
                else v[r] += bn;
            }
            if (full) {
                // (round 4: all sixteen residual values requested before the first store -- out may alias resid, so hipcc kept
                // every load behind the previous store and waited for both: 16 dependent L2 round trips per 32 x 32 block)
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += rp[(long long)((r & 3) + 8 * (r >> 2)) * p.N];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) op[(long long)((r & 3) + 8 * (r >> 2)) * p.N] = v[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (mb + dr < p.M) {
                        const long long off = (long long)dr * p.N;
                        op[off] = (EPI == EPI_RESID) ? v[r] + rp[off] : v[r];
                    }
                }
            }
        }
    }
}

template <int BM, int BN>
constexpr size_t gemm_lds_bytes() { return (size_t)2 * (BM + BN) * kRowBytes; }

// DW = 1 builds the variant with a 5th, dedicated LDS-DMA wave.  Measured NEGATIVE on MI355X (tools/
// gemm_lab: s2.pw1 657 vs 581 us, s2.pw2 742 vs 627 us): one wave cannot issue the 32 pieces of a tile and see
// them land within one k-tile of MFMA time, so it becomes the critical path; 8 pieces on each MFMA wave is faster.
template <int kBM, int BN, int WM, int WN, int EPI, int GATHER>
static int launch_cfg(const GemmParams& p0, hipStream_t s) {
    constexpr int DW = 0;
    GemmParams p = p0;
    p.tiles_n = p.N / BN;
    const long long tiles_m = (p.M + kBM - 1) / kBM;
    const long long blocks = tiles_m * p.tiles_n;
    if (blocks > 0x7fffffffLL) ACX_FAIL(ACX_ERR_SHAPE, "gemm: grid too large");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &gemm_f32_kernel<kBM, BN, WM, WN, EPI, GATHER, DW>, gemm_lds_bytes<kBM, BN>()));
    constexpr size_t lds = gemm_lds_bytes<kBM, BN>();
    launch_kernel(&gemm_f32_kernel<kBM, BN, WM, WN, EPI, GATHER, DW>, dim3((unsigned)blocks), dim3(256 + 64 * DW), lds, s, p);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

template <int EPI, int GATHER>
static int launch_bn(const GemmParams& p, hipStream_t s) {
    // 128-row tiles by default; 64-row tiles when the 128-row grid would not fill ~1.5 rounds of the
    // 512 resident workgroups (stage-3 pwconv2, the last downsample): halves the tail-round loss.
    const long long tiles128 = ((p.M + 127) / 128) * (p.N % 128 == 0 ? p.N / 128 : p.N / 96);
    const bool small = tiles128 < 800;
    if (p.N % 128 == 0) {
        if (small) return launch_cfg<64, 128, 2, 2, EPI, GATHER>(p, s);
        return launch_cfg<128, 128, 2, 2, EPI, GATHER>(p, s);
    }
    if (p.N % 96 == 0) return launch_cfg<128, 96, 4, 1, EPI, GATHER>(p, s);
    ACX_FAIL(ACX_ERR_SHAPE, "gemm: N=%d is not a multiple of 96 or 128", p.N);
}

int launch_gemm(acx_ctx* c, const GemmArgs& a, hipStream_t s) {
    if (a.K % kBK != 0) ACX_FAIL(ACX_ERR_SHAPE, "gemm: K=%d is not a multiple of %d", a.K, kBK);
    if (a.M <= 0) return ACX_OK;
    GemmParams p;
    p.A = a.A; p.Wt = a.Wt; p.bias = a.bias; p.out = a.out; p.stats = a.stats; p.colsum = a.colsum; p.resid = a.resid;
    p.M = a.M; p.N = a.N; p.K = a.K; p.H = a.H; p.W = a.W; p.C = a.C; p.Ho = a.Ho; p.Wo = a.Wo;
    p.tiles_n = 0;
    ProfScope ps(c, a.cls, s);
    if (a.gather) {
        if (a.epi != EPI_BIAS || a.C % kBK != 0) ACX_FAIL(ACX_ERR_ARG, "gemm: bad gather configuration");
        return launch_bn<EPI_BIAS, 1>(p, s);
    }
    if (a.epi == EPI_GELU) {
        if (!a.stats || !a.colsum) ACX_FAIL(ACX_ERR_ARG, "gemm: the LayerNorm+GELU epilogue needs row stats and column sums");
        return launch_bn<EPI_GELU, 0>(p, s);
    }
    if (a.epi == EPI_RESID) return launch_bn<EPI_RESID, 0>(p, s);
    if (a.epi == EPI_BIAS) return launch_bn<EPI_BIAS, 0>(p, s);
    ACX_FAIL(ACX_ERR_ARG, "gemm: unknown epilogue %d", a.epi);
}

}  // namespace acx
