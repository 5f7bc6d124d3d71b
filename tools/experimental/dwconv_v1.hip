// K3 -- depthwise 7x7 convolution, zero pad 3, + bias, channels-last (Block.dwconv,
// convnext.py:58-60 called at :76), and the per-pixel LayerNorm statistics that Block.norm
// (convnext.py:61,78; F.layer_norm :532-535) needs.
//
// HBM-bound: algorithmic bytes = read x + write y = 2*C*H*W*4 B per clip per block
// (10.84 / 5.42 / 2.71 / 1.33 MB in stages 0-3).  12.25 FLOP/B, so the VALU must stay under
// ~50 % busy to reach the HBM roofline: lanes run along C (float4 = 16 B per lane, 8 lanes = one
// 128-B line per pixel), each thread keeps WT=7 adjacent output pixels x 4 channels in registers
// and slides the 7-tap row over 13 LDS reads (49 FMA x 4 channels per 13+7 ds_read_b128).
// The input halo tile and the 49x32 weight slice are staged through LDS once per workgroup.
#include "../../audioset-convnext-inf_amd/csrc/acx_internal.h"

namespace acx {

constexpr int kDwSlice = 32;      // channels per workgroup (8 lanes x float4)

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef ACX_LAB_DW_STAMP      // diagnostic build (tools/dw_lab.hip): where does a tile spend its cycles?
__device__ unsigned long long acx_dw_stamps[8];
#define ACX_STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define ACX_STAMP(var)
#endif

// Streaming form: a workgroup owns (clip, 32-channel slice, column strip of TW pixels, segment of row tiles)
// and walks DOWN the image.  An LDS ring of TH+6 input rows is kept; each step computes TH output rows from
// the ring while the next TH input rows are already in flight to registers (issued before the FMAs, written
// into the ring slots of the TH oldest rows after them), so every input row is fetched once per strip
// (the first tile-only version re-read its 6 halo rows per tile: FETCH_SIZE 1.93x the algorithmic bytes,
// profiles/r01_c_pmc_per_kernel.csv) and HBM latency hides under the arithmetic.
template <int TW, int TH>
struct DwCfg {
    static constexpr int WT = 7;
    static constexpr int kStrips = TW / WT;
    static constexpr int kThreads = kStrips * 8 * TH;
    static constexpr int kCols = TW + 6;
    static constexpr int kRing = TH + 6;
    static constexpr int kRowF4 = kCols * 8;                      // float4 per ring row
    static constexpr int kStepF4 = TH * kRowF4;                   // float4 fetched per step
    static constexpr int kStage = (kStepF4 + kThreads - 1) / kThreads;   // staging float4 per thread
    static constexpr size_t kLdsBytes = (size_t)(kRing * kRowF4 + 49 * 8 + 256) * 16;   // ring + weights + sinks
};

template <int TW, int TH>
__global__ __launch_bounds__(256, 2) void dwconv7_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ wt /*[49][C]*/,
                                                         const float* __restrict__ bias, int H, int W, int C,
                                                         int tiles_w, int tiles_h, int n_seg) {
    using Cfg = DwCfg<TW, TH>;
    static_assert(Cfg::kThreads == 256, "thread mapping assumes 256 threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* ring = reinterpret_cast<f32x4*>(smem);                  // [kRing][kCols][8]
    f32x4* wl = ring + Cfg::kRing * Cfg::kRowF4;                   // [49][8]
    f32x4* dummy = wl + 49 * 8;                                    // [256] per-thread sink for out-of-image columns

    int bid = blockIdx.x;
    const int slice = bid % (C / kDwSlice); bid /= (C / kDwSlice);
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int seg = bid % n_seg; bid /= n_seg;
    const long long b = bid;
    const int c0 = slice * kDwSlice;
    const int w0 = tw * TW;
    const int tid = threadIdx.x;
    const int t_begin = (int)((long long)tiles_h * seg / n_seg);     // balanced segments of row tiles
    const int t_end = (int)((long long)tiles_h * (seg + 1) / n_seg);
    if (t_begin >= t_end) return;

    for (int i = tid; i < 49 * 8; i += Cfg::kThreads)
        wl[i] = *reinterpret_cast<const f32x4*>(wt + (i >> 3) * C + c0 + 4 * (i & 7));
    // columns outside the image are never written again: zero the whole ring once
    for (int i = tid; i < Cfg::kRing * Cfg::kRowF4; i += Cfg::kThreads) ring[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xb = x + b * (long long)H * W * C + c0;
    const long long row_elems = (long long)W * C;

    // Per-thread staging plan for a step of TH image rows (float4 #k of this thread):
    //   st_ptr[k]  source of (row st_row[k], column, quad) for image row 0   (columns clamped into the image)
    //   st_lds[k]  float4 index inside a ring row, or -1 - (dummy index) when the column is outside the image
    // Loads are UNCONDITIONAL on valid addresses (a per-element "load or zero" makes hipcc branch around every
    // load and wait for all of them on the spot); rows outside the image only occur in the first/last step of a
    // column and take the slow (clamp + zero) path, selected by a wave-uniform test.
    const float* st_ptr[Cfg::kStage];
    int st_row[Cfg::kStage], st_lds[Cfg::kStage];
#pragma unroll
    for (int k = 0; k < Cfg::kStage; ++k) {
        int i = tid + k * Cfg::kThreads;
        const bool in_step = i < Cfg::kStepF4;
        if (!in_step) i = Cfg::kStepF4 - 1;
        const int qq = i & 7;
        const int col = (i >> 3) % Cfg::kCols;
        st_row[k] = (i >> 3) / Cfg::kCols;
        const int gw = w0 - 3 + col;
        const int gwc = gw < 0 ? 0 : (gw >= W ? W - 1 : gw);
        st_ptr[k] = xb + st_row[k] * row_elems + gwc * C + 4 * qq;
        st_lds[k] = (in_step && gw >= 0 && gw < W) ? (col * 8 + qq) : -1;
    }
    const int g_origin = t_begin * TH - 3;             // image row kept in ring row 0 of this segment
    f32x4 rg[Cfg::kStage];
#define ACX_DW_LOAD_FAST(first_row)   /* rows first_row .. +TH-1, all inside the image */                    \
    {                                                                                                     \
        const long long roff = (long long)(first_row) * row_elems;                                        \
        _Pragma("unroll") for (int k = 0; k < Cfg::kStage; ++k)                                           \
            rg[k] = *reinterpret_cast<const f32x4*>(st_ptr[k] + roff);                                    \
    }
#define ACX_DW_LOAD_EDGE(first_row)   /* some rows outside [0,H): clamp the row, zero at store time */      \
    _Pragma("unroll") for (int k = 0; k < Cfg::kStage; ++k) {                                             \
        int gh = (first_row) + st_row[k];                                                                 \
        gh = gh < 0 ? 0 : (gh >= H ? H - 1 : gh);                                                         \
        rg[k] = *reinterpret_cast<const f32x4*>(st_ptr[k] + (long long)(gh - st_row[k]) * row_elems);     \
    }
#define ACX_DW_STORE(first_row, max_rows, edge) /* registers -> ring rows of image rows first_row .. */     \
    {                                                                                                     \
        const int slot0 = ((first_row) - g_origin) % Cfg::kRing;           /* wave-uniform */             \
        _Pragma("unroll") for (int k = 0; k < Cfg::kStage; ++k) {                                         \
            int slot = slot0 + st_row[k];                                                                 \
            if (slot >= Cfg::kRing) slot -= Cfg::kRing;                                                   \
            f32x4 v = rg[k];                                                                              \
            bool keep = st_lds[k] >= 0 && st_row[k] < (max_rows);                                         \
            if (edge) {                                                                                   \
                const int gh = (first_row) + st_row[k];                                                   \
                if (gh < 0 || gh >= H) v = f32x4{0.f, 0.f, 0.f, 0.f};                                     \
            }                                                                                             \
            f32x4* dst = keep ? ring + slot * Cfg::kRowF4 + st_lds[k] : dummy + tid;                      \
            *dst = v;                                                                                     \
        }                                                                                                 \
    }
    __syncthreads();                                   // ring zeroed before the prologue fills it
    // prologue: image rows [g_origin, g_origin + TH + 6) in two rounds (of the second only 6 rows are kept)
    ACX_DW_LOAD_EDGE(g_origin)
    ACX_DW_STORE(g_origin, TH, true)
    ACX_DW_LOAD_EDGE(g_origin + TH)
    ACX_DW_STORE(g_origin + TH, 6, true)
    __syncthreads();

    const int q = tid & 7;
    const int strip = (tid >> 3) % Cfg::kStrips;
    const int r = (tid >> 3) / Cfg::kStrips;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c0 + 4 * q);
    const int rd_off = (strip * Cfg::WT) * 8 + q;      // float4 offset of this thread's first input column
    float* const yb = y + ((b * H) * (long long)W + w0 + strip * Cfg::WT) * C + c0 + 4 * q;

#ifdef ACX_LAB_DW_STAMP
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, acc_s[5] = {0, 0, 0, 0, 0};
#endif
    f32x4 so[Cfg::WT];
#pragma unroll
    for (int i = 0; i < Cfg::WT; ++i) so[i] = bv;
    float* yp_prev = nullptr;                          // where so[] belongs (null: nothing pending)
#ifdef ACX_LAB_DW_NOSTORE
#define ACX_DW_FLUSH if (yp_prev != nullptr && so[0][0] == 12345.678f) { _Pragma("unroll") for (int i = 0; i < Cfg::WT; ++i) *reinterpret_cast<f32x4*>(yp_prev + (long long)i * C) = so[i]; }
#else
#define ACX_DW_FLUSH if (yp_prev != nullptr) { _Pragma("unroll") for (int i = 0; i < Cfg::WT; ++i) *reinterpret_cast<f32x4*>(yp_prev + (long long)i * C) = so[i]; }
#endif
    for (int t = t_begin; t < t_end; ++t) {
        const int h0 = t * TH;
        const bool more = t + 1 < t_end;
        ACX_STAMP(ts0)
        const int next_first = h0 + TH + 3;            // image rows of the next step (ring rows of the TH oldest)
        const bool edge = next_first + TH > H;         // (next_first >= 0 always)
#ifndef ACX_LAB_DW_NOLOAD
        if (more) {
            if (edge) { ACX_DW_LOAD_EDGE(next_first) } else { ACX_DW_LOAD_FAST(next_first) }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // Outputs leave one tile LATE, from their own register set, right BEHIND the next prefetch: hipcc puts a
        // vmcnt(0) in front of the prefetch address set-up (it cannot prove the staging registers idle across
        // the back edge), so nothing may be in flight there; the one real wait of the loop (before the ring
        // refill) then covers loads and stores that both had a whole FMA phase to complete.
        ACX_DW_FLUSH
        __builtin_amdgcn_sched_barrier(0);

        f32x4 acc[Cfg::WT];
#pragma unroll
        for (int i = 0; i < Cfg::WT; ++i) acc[i] = bv;
        int qw = q;                       // opaque per tile: keeps hipcc from hoisting all 49 weight float4
        asm volatile("" : "+v"(qw));      // (196 VGPRs) out of the tile loop -- they are re-read from LDS instead
        const int base = (h0 - 3 - g_origin + r) % Cfg::kRing;    // ring slot of input row h0 - 3 + r
        // The 7 kernel rows are software-pipelined by hand in HALF rows: the LDS reads of the next unit
        // (7 or 6 input float4, and once per row the 7 weight float4 of the next kernel row) are issued into
        // a second register set BEFORE the 49 packed FMAs of the current unit.  A wave issues at most one
        // instruction per ~5 cycles and only two waves fit per SIMD, so every non-FMA instruction costs FMA
        // time: one wait per unit (single asm touch), no per-element address math.
        f32x4 iA[7], iB[6], wA[7], wB[7];
#define ACX_DW_ROWP(ky_, p_)                                                                              \
        const f32x4* p_;                                                                                  \
        {                                                                                                 \
            int slot = base + (ky_);                                                                      \
            if (slot >= Cfg::kRing) slot -= Cfg::kRing;                                                   \
            p_ = ring + slot * Cfg::kRowF4 + rd_off;                                                      \
        }
#define ACX_DW_READ_W(w_, ky_) _Pragma("unroll") for (int kx = 0; kx < 7; ++kx) w_[kx] = wl[((ky_) * 7 + kx) * 8 + qw];
#define ACX_DW_READ_I0(ky_) { ACX_DW_ROWP(ky_, p0_) _Pragma("unroll") for (int j = 0; j < 7; ++j) iA[j] = p0_[j * 8]; }
#define ACX_DW_READ_I1(ky_) { ACX_DW_ROWP(ky_, p1_) _Pragma("unroll") for (int j = 0; j < 6; ++j) iB[j] = p1_[(7 + j) * 8]; }
#ifdef ACX_LAB_DW_SCALAR_FMA
#define ACX_DW_MAC(a_, i_, w_)                                                                             \
        {                                                                                                 \
            float t0, t1, t2, t3;                                                                         \
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(i_[0]), "v"(w_[0]), "v"(a_[0]));     \
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(i_[1]), "v"(w_[1]), "v"(a_[1]));     \
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t2) : "v"(i_[2]), "v"(w_[2]), "v"(a_[2]));     \
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(i_[3]), "v"(w_[3]), "v"(a_[3]));     \
            a_[0] = t0; a_[1] = t1; a_[2] = t2; a_[3] = t3;                                               \
        }
#else
#define ACX_DW_MAC(a_, i_, w_) a_ += i_ * w_;
#endif
        // opaque re-definition of acc pins the FMAs in place (plain arithmetic is otherwise sunk below later reads)
#define ACX_DW_PIN asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]));
#define ACX_DW_FMA0(w_)                                                                                   \
        {                                                                                                 \
            _Pragma("unroll") for (int j = 0; j < 7; ++j)                                                 \
            _Pragma("unroll") for (int kx = 0; kx <= j; ++kx) ACX_DW_MAC(acc[j - kx], iA[j], w_[kx])      \
            ACX_DW_PIN                                                                                    \
        }
#define ACX_DW_FMA1(w_)                                                                                   \
        {                                                                                                 \
            _Pragma("unroll") for (int j = 7; j < 13; ++j)                                                \
            _Pragma("unroll") for (int kx = j - 6; kx < 7; ++kx) ACX_DW_MAC(acc[j - kx], iB[j - 7], w_[kx]) \
            ACX_DW_PIN                                                                                    \
        }
        // one asm statement per register set: ONE lgkmcnt wait, placed in front of the next batch of reads
#define ACX_DW_TOUCH6(a_) asm volatile("" :: "v"(a_[0]), "v"(a_[1]), "v"(a_[2]), "v"(a_[3]), "v"(a_[4]), "v"(a_[5]));
#define ACX_DW_TOUCH7(a_) asm volatile("" :: "v"(a_[0]), "v"(a_[1]), "v"(a_[2]), "v"(a_[3]), "v"(a_[4]), "v"(a_[5]), "v"(a_[6]));
        // LDS reads are SPREAD between the FMAs (sched_group_barrier: 1 ds_read, then a few VALU, repeated):
        // eight waves bursting 14 reads each overflow the LDS queue and stall the issuing waves.
#define ACX_DW_MIX(nread_, nvalu_)                                                                        \
        _Pragma("unroll") for (int z = 0; z < (nread_); ++z) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                            \
            __builtin_amdgcn_sched_group_barrier(0x002, (nvalu_), 0);                                     \
        }
#define ACX_DW_KROW(ky_, wc_, wn_)   /* kernel row ky_ with weights wc_; prefetches row ky_+1 into wn_ */   \
        {                                                                                                 \
            ACX_DW_READ_I1(ky_)                                                                           \
            ACX_DW_FMA0(wc_)                                                                              \
            ACX_DW_MIX(6, 9)                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            ACX_DW_TOUCH6(iB)                                                                             \
            if ((ky_) + 1 < 7) { ACX_DW_READ_W(wn_, (ky_) + 1) ACX_DW_READ_I0((ky_) + 1) }                \
            ACX_DW_FMA1(wc_)                                                                              \
            if ((ky_) + 1 < 7) { ACX_DW_MIX(14, 3) }                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if ((ky_) + 1 < 7) { ACX_DW_TOUCH7(wn_) ACX_DW_TOUCH7(iA) }                                   \
        }
        ACX_DW_READ_W(wA, 0)
        ACX_DW_READ_I0(0)
#ifndef ACX_LAB_DW_NOFMA
        ACX_DW_KROW(0, wA, wB) ACX_DW_KROW(1, wB, wA) ACX_DW_KROW(2, wA, wB) ACX_DW_KROW(3, wB, wA)
        ACX_DW_KROW(4, wA, wB) ACX_DW_KROW(5, wB, wA) ACX_DW_KROW(6, wA, wB)
#endif
#undef ACX_DW_ROWP
#undef ACX_DW_READ_W
#undef ACX_DW_READ_I0
#undef ACX_DW_READ_I1
#undef ACX_DW_PIN
#undef ACX_DW_MAC
#undef ACX_DW_FMA0
#undef ACX_DW_FMA1
#undef ACX_DW_TOUCH6
#undef ACX_DW_TOUCH7
#undef ACX_DW_KROW
#undef ACX_DW_MIX
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
#ifdef ACX_LAB_DW_STAMP
            acc_s[0] += ts1 - ts0; acc_s[1] += ts2 - ts1; acc_s[2] += ts3 - ts2; acc_s[3] += ts4 - ts3; acc_s[4] += ts5 - ts4;
#endif
        }
#pragma unroll
        for (int i = 0; i < Cfg::WT; ++i) so[i] = acc[i];
        yp_prev = (h0 + r < H) ? yb + (long long)(h0 + r) * row_elems : nullptr;
    }
    ACX_DW_FLUSH
#undef ACX_DW_FLUSH
#ifdef ACX_LAB_DW_STAMP
    if ((tid & 63) == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&acx_dw_stamps[i], acc_s[i]);
        atomicAdd(&acx_dw_stamps[5], (unsigned long long)(t_end - t_begin - 1));
    }
#endif
#undef ACX_DW_LOAD_FAST
#undef ACX_DW_LOAD_EDGE
#undef ACX_DW_STORE
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
        case 96: rowstats_kernel<8, NORMALIZE><<<grid, blk, 0, s>>>(x, out, M, 1e-6f); break;
        case 192: rowstats_kernel<16, NORMALIZE><<<grid, blk, 0, s>>>(x, out, M, 1e-6f); break;
        case 384: rowstats_kernel<32, NORMALIZE><<<grid, blk, 0, s>>>(x, out, M, 1e-6f); break;
        case 768: rowstats_kernel<64, NORMALIZE><<<grid, blk, 0, s>>>(x, out, M, 1e-6f); break;
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

template <int TW, int TH>
static int launch_dw_cfg(const BlockW& w, int C, const float* x, float* y, int B, int H, int W, hipStream_t s) {
    using Cfg = DwCfg<TW, TH>;
    static bool attr_set = false;
    if (!attr_set) {
        ACX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv7_kernel<TW, TH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::kLdsBytes));
        attr_set = true;
    }
    const int tiles_w = W / TW, tiles_h = (H + TH - 1) / TH;
    const long long columns = (long long)B * tiles_w * (C / kDwSlice);
    // Workgroups = columns x row segments.  Two workgroups are resident per CU (512 slots): aim at a whole
    // number of rounds (3 x 512) -- a fractional last round idles half the chip for a whole workgroup life
    // (measured 4.5 rounds = 5) -- with segments of at least 2 row tiles to amortise the ring prologue.
    // (stage 2, 768 columns: ONE segment of 4 tiles per workgroup beats two of 2 -- 61 vs 73 us in tools/dw_lab --
    //  although 768 workgroups are only 1.5 rounds: the ring prologue is the larger cost)
    int n_seg = columns >= 768 ? 1 : (int)((1536 + columns / 2) / columns);
    if (n_seg > tiles_h / 2) n_seg = tiles_h / 2;
    if (n_seg < 1) n_seg = 1;
#ifdef ACX_LAB_DW_NSEG       // diagnostic (tools/dw_lab.hip): force the number of row segments
    n_seg = ACX_LAB_DW_NSEG > tiles_h ? tiles_h : ACX_LAB_DW_NSEG;
#endif
    const long long blocks = columns * n_seg;
    dwconv7_kernel<TW, TH><<<dim3((unsigned)blocks), dim3(Cfg::kThreads), Cfg::kLdsBytes, s>>>(
        x, y, w.dw, w.dwb, H, W, C, tiles_w, tiles_h, n_seg);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

int launch_dwconv(acx_ctx* c, const BlockW& w, int C, const float* x, float* y, float* stats, int B, int H,
                  int W, hipStream_t s) {
    {
        ProfScope ps(c, ACX_K_DWCONV, s);
        int rc;
        switch (W) {
            case 56: rc = launch_dw_cfg<28, 8>(w, C, x, y, B, H, W, s); break;
            case 28: rc = launch_dw_cfg<28, 8>(w, C, x, y, B, H, W, s); break;
            case 14: rc = launch_dw_cfg<14, 16>(w, C, x, y, B, H, W, s); break;
            case 7: rc = launch_dw_cfg<7, 32>(w, C, x, y, B, H, W, s); break;   // 31x7 image: one step
            default: ACX_FAIL(ACX_ERR_SHAPE, "dwconv7: unsupported width %d (expected 56/28/14/7)", W);
        }
        ACX_TRY(rc);
    }
    if (stats) ACX_TRY(launch_rowstats(c, y, stats, (int64_t)B * H * W, C, s));
    return ACX_OK;
}

}  // namespace acx
