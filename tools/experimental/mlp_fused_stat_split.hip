// EXPERIMENTAL, NOT BUILT INTO libacx (round 2): correct (parity + stress suites green, 107 tests) but NOT faster than the
// kernel that ships for stage 0 -- 2 x 295 us (LNOUT second launch 347) per block against 596-638 us for the 8-wave ring
// kernel mlp_fused_split_kernel<96> on the same box (bench: 6 232-6 277 vs 6 282 clips/s).  Each half launch streams the
// whole y and x again (1.04 GB): ~210 us of memory phase per launch that its ~200 us of matrix + vector issue only partly
// cover.  To build it: paste into mlp_fused_wide.hip (it uses WideCfg<96, 1>, gelu_micro and the file's -fno-slp-vectorize
// build), declare it in acx_internal.h, pack BlockW::wstream_h in api.hip with segment order
//   W1(k) -> (k / 6) * 12 + k % 6,   W2(k) -> (k / 6) * 12 + 6 + k % 6      (images as for mlp_fused_wide_kernel)
// and call launch_mlp_fused_stat_split from run_block.
//
// ---- C = 96: weights-stationary variant, one HALF of the hidden units per launch ---------------------------------------
// At C = 96 a workgroup cannot keep the block's weights (2 x 4C x C S16 values = 288 KB) in the LDS, so the kernels above
// stream them per pixel tile -- LDS-DMA issues and a barrier per segment of only 18 MFMAs, all waves in step.  HALF of the
// hidden units (192 of 384: 6 chunks, 12 segments, 144 KB) does fit.  The block therefore runs as TWO launches,
//   launch h:   x += gamma W2[:, half h] . GELU(W1[half h] . LN(y) + b1[half h])   (+ gamma b2 in launch 0)
// each by one persistent CU-exclusive workgroup per CU that loads its 12 segments once; after that its 8 waves never
// synchronise again: each walks its own 32-pixel tiles (LayerNorm -> 6 chunks of [phase 1, GELU + split, phase 2] ->
// read-modify-write of x) at its own pace.  LayerNorm(y) is computed in both launches, x is read and written twice
// (1.04 GB more traffic per block, mostly from the Infinity Cache) -- cheaper than the per-segment costs it removes.
// Same arithmetic per product as the kernels above; the sum over the hidden units is split into two fp32 partial sums.
// The second launch of the last block of a stage writes the downsample GEMM's S16 operand rows instead of x (LNOUT).
template <bool LNOUT>
__global__ __launch_bounds__(512) void mlp_fused_stat_split_kernel(
    const float* __restrict__ y, float* __restrict__ x, const char* __restrict__ wstream /*[12][12288 B]: W1(0..5) W2(0..5) of this half*/,
    const float* __restrict__ b1 /*[192] of this half*/, const float* __restrict__ b2 /*[96] or nullptr (second launch)*/,
    long long M, float sinv1, float sinv2, float hscale, char* __restrict__ ln_out) {
    constexpr int C = 96;
    using Cfg = WideCfg<C, 1>;
    constexpr int kHalfChunks = Cfg::kChunks / 2;                  // 6 chunks of 32 hidden units
    constexpr int kStreamBytes = 2 * kHalfChunks * Cfg::kSegBytes; // 147 456
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + kStreamBytes);    // [192], pre-divided by sinv1
    float* b2s = b1s + 32 * kHalfChunks;                           // [96] (zeros in the second launch)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    ACX_CLAIM_VGPR(255);          // CU-exclusive: 8 waves x 256 registers hold the SIMDs' whole register files

#pragma unroll
    for (int p = 0; p < kStreamBytes / 1024 / 8; ++p) {
        const int piece = wave * (kStreamBytes / 1024 / 8) + p;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wstream + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, 0, 0);
    }
    {
        const float b1scale = 1.0f / sinv1;             // a power of two
        if (tid < 32 * kHalfChunks) b1s[tid] = b1[tid] * b1scale;
        if (tid < C) b2s[tid] = b2 ? b2[tid] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int w1off[4][2], w2off[2][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w1off[q][pl] = l31 * (4 * C) + (((4 * q + 2 * hh + pl) ^ Cfg::swz1(l31)) << 4);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w2off[sp][pl] = l31 * 128 + (((2 * (2 * sp + hh) + pl) ^ ((l31 >> 1) & 7)) << 4);
    GeluConsts gk;
    gk.ps = 0.3275911f * 0.70710678f * sinv1;
    gk.cq = 0.84932180f * sinv1;
    gk.ca = -0.5f * sinv1 * hscale;
    gk.cb = sinv1 * hscale;
#define ACX_H8(v_) __builtin_bit_cast(h8, v_)
#define ACX_W1_RD(base_, s_, pl_) (*reinterpret_cast<const f32x4*>((base_) + ((s_) >> 2) * 256 + w1off[(s_) & 3][pl_]))
#define ACX_W2_RD(base_, i_, pl_) (*reinterpret_cast<const f32x4*>((base_) + ((i_) >> 1) * 4096 + w2off[(i_) & 1][pl_]))

    const long long ntiles = (M + 31) / 32;
    for (long long tile = (long long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long long)gridDim.x * 8) {
        long long mrow = tile * 32 + l31;
        const bool valid = mrow < M;
        if (!valid) mrow = M - 1;
        // ---- LayerNorm of the tile's rows -> S16 halves: lane (px = l31, half hh) holds channels 16 s + 8 hh .. + 7 ----
        f32x4 acth[Cfg::kSteps], actl[Cfg::kSteps];
        {
            float a[C / 2];
            const float* yp = y + mrow * C + 8 * hh;
#pragma unroll
            for (int s = 0; s < Cfg::kSteps; ++s) {
                const float4 v0 = *reinterpret_cast<const float4*>(yp + 16 * s);
                const float4 v1 = *reinterpret_cast<const float4*>(yp + 16 * s + 4);
                a[8 * s + 0] = v0.x; a[8 * s + 1] = v0.y; a[8 * s + 2] = v0.z; a[8 * s + 3] = v0.w;
                a[8 * s + 4] = v1.x; a[8 * s + 5] = v1.y; a[8 * s + 6] = v1.z; a[8 * s + 7] = v1.w;
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
                unsigned uh4[4], ul4[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f32x2 v;
                    v.x = (a[8 * s + 2 * p] - mean) * sc; v.y = (a[8 * s + 2 * p + 1] - mean) * sc;
                    const h2 h = __builtin_convertvector(v, h2);
                    const f32x2 back = __builtin_convertvector(h, f32x2);
                    const h2 l = __builtin_convertvector(v - back, h2);
                    uh4[p] = __builtin_bit_cast(unsigned, h);
                    ul4[p] = __builtin_bit_cast(unsigned, l);
                }
                acth[s] = __builtin_bit_cast(f32x4, uint4{uh4[0], uh4[1], uh4[2], uh4[3]});
                actl[s] = __builtin_bit_cast(f32x4, uint4{ul4[0], ul4[1], ul4[2], ul4[3]});
            }
        }
        f32x16 acc[C / 32];
#pragma unroll
        for (int t = 0; t < C / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#pragma nounroll
        for (int k = 0; k < kHalfChunks; ++k) {
            const char* base1 = smem + k * Cfg::kSegBytes;
            const char* base2 = smem + (kHalfChunks + k) * Cfg::kSegBytes;
            f32x16 X;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + 32 * k + 8 * q + 4 * hh);
                X[4 * q + 0] = bq[0]; X[4 * q + 1] = bq[1]; X[4 * q + 2] = bq[2]; X[4 * q + 3] = bq[3];
            }
#pragma unroll
            for (int s = 0; s < Cfg::kSteps; ++s) {
                const f32x4 ah = ACX_W1_RD(base1, s, 0), al = ACX_W1_RD(base1, s, 1);
                X = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al), ACX_H8(acth[s]), X, 0, 0, 0);
                X = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah), ACX_H8(actl[s]), X, 0, 0, 0);
                X = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah), ACX_H8(acth[s]), X, 0, 0, 0);
            }
            // GELU + hi / lo split: pairs 0..3 -> k-step 0 of phase 2, pairs 4..7 -> k-step 1
            f32x4 gh[2], gl[2];
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                unsigned uh[4], ul[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    GeluState gs;
                    gs.ax = X[2 * (4 * sp + p)]; gs.ay = X[2 * (4 * sp + p) + 1];
                    gelu_micro<0>(gs, gk, uh[p], ul[p]); gelu_micro<1>(gs, gk, uh[p], ul[p]); gelu_micro<2>(gs, gk, uh[p], ul[p]);
                    gelu_micro<3>(gs, gk, uh[p], ul[p]); gelu_micro<4>(gs, gk, uh[p], ul[p]); gelu_micro<5>(gs, gk, uh[p], ul[p]);
                    gelu_micro<6>(gs, gk, uh[p], ul[p]); gelu_micro<7>(gs, gk, uh[p], ul[p]); gelu_micro<8>(gs, gk, uh[p], ul[p]);
                }
                gh[sp] = __builtin_bit_cast(f32x4, uint4{uh[0], uh[1], uh[2], uh[3]});
                gl[sp] = __builtin_bit_cast(f32x4, uint4{ul[0], ul[1], ul[2], ul[3]});
            }
#pragma unroll
            for (int i = 0; i < Cfg::kUnits; ++i) {
                const f32x4 ah = ACX_W2_RD(base2, i, 0), al = ACX_W2_RD(base2, i, 1);
                acc[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(al), ACX_H8(gh[i & 1]), acc[i >> 1], 0, 0, 0);
                acc[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah), ACX_H8(gl[i & 1]), acc[i >> 1], 0, 0, 0);
                acc[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ACX_H8(ah), ACX_H8(gh[i & 1]), acc[i >> 1], 0, 0, 0);
            }
        }

        // ---- epilogue: lane (px, hh), tile t, q: channels 32 t + 8 q + 4 hh .. + 3  ->  x += out (+ b2) ----
        const float* xp = x + mrow * C + 4 * hh;
        if constexpr (LNOUT) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < C / 32; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * t + 8 * q;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + c + 4 * hh);
                    const float4 v = *reinterpret_cast<const float4*>(xp + c);
                    acc[t][4 * q + 0] = v.x + fmaf(acc[t][4 * q + 0], sinv2, bb.x);
                    acc[t][4 * q + 1] = v.y + fmaf(acc[t][4 * q + 1], sinv2, bb.y);
                    acc[t][4 * q + 2] = v.z + fmaf(acc[t][4 * q + 2], sinv2, bb.z);
                    acc[t][4 * q + 3] = v.w + fmaf(acc[t][4 * q + 3], sinv2, bb.w);
                    sum += (acc[t][4 * q + 0] + acc[t][4 * q + 1]) + (acc[t][4 * q + 2] + acc[t][4 * q + 3]);
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
            if (valid) {
                char* op = ln_out + mrow * (long long)(C * 4) + 8 * hh;
#pragma unroll
                for (int t = 0; t < C / 32; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned uhi[2], ulo[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            f32x2 v;
                            v.x = (acc[t][4 * q + 2 * e] - mean) * sc; v.y = (acc[t][4 * q + 2 * e + 1] - mean) * sc;
                            const h2 h = __builtin_convertvector(v, h2);
                            const f32x2 back = __builtin_convertvector(h, f32x2);
                            const h2 l = __builtin_convertvector(v - back, h2);
                            uhi[e] = __builtin_bit_cast(unsigned, h);
                            ulo[e] = __builtin_bit_cast(unsigned, l);
                        }
                        char* blk = op + (4 * t + q) * 32;
                        *reinterpret_cast<uint2*>(blk) = uint2{uhi[0], uhi[1]};
                        *reinterpret_cast<uint2*>(blk + 16) = uint2{ulo[0], ulo[1]};
                    }
            }
        } else if (valid) {
            float* xo = x + mrow * C + 4 * hh;
#pragma unroll
            for (int t = 0; t < C / 32; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * t + 8 * q;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b2s + c + 4 * hh);
                    float4 v = *reinterpret_cast<const float4*>(xp + c);
                    v.x += fmaf(acc[t][4 * q + 0], sinv2, bb.x);
                    v.y += fmaf(acc[t][4 * q + 1], sinv2, bb.y);
                    v.z += fmaf(acc[t][4 * q + 2], sinv2, bb.z);
                    v.w += fmaf(acc[t][4 * q + 3], sinv2, bb.w);
                    *reinterpret_cast<float4*>(xo + c) = v;
                }
        }
    }
#undef ACX_H8
#undef ACX_W1_RD
#undef ACX_W2_RD
}

template <bool LNOUT>
static int launch_stat_split(const BlockW& w, int half, const float* y, float* x, long long M, void* ln_out, hipStream_t s) {
    static_assert(WideCfg<96, 1>::kSegs / 2 * WideCfg<96, 1>::kSegBytes + (192 + 96) * 4 <= kCuLdsBytes, "half the stream does not fit the LDS");
    static DeviceOnce once;
    ACX_TRY(set_max_dynamic_lds(once, &mlp_fused_stat_split_kernel<LNOUT>, kCuLdsBytes));
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long long wgs_needed = ((M + 31) / 32 + 7) / 8;
    const long long blocks = wgs_needed < cus ? wgs_needed : cus;
    const char* stream = reinterpret_cast<const char*>(w.wstream_h) + (size_t)half * (WideCfg<96, 1>::kSegs / 2) * WideCfg<96, 1>::kSegBytes;
    mlp_fused_stat_split_kernel<LNOUT><<<dim3((unsigned)blocks), dim3(512), kCuLdsBytes /* CU-exclusive */, s>>>(
        y, x, stream, w.b1 + 192 * half, half == 0 ? w.b2 : nullptr, M, 1.0f / (kSplitLnScale * w.w1s_scale),
        1.0f / (w.hid_scale * w.w2s_scale), w.hid_scale, reinterpret_cast<char*>(ln_out));
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

bool mlp_fused_stat_split_supported(int C) { return C == 96; }

// both launches of a block (see the kernel); ln_out: the second one writes the downsample GEMM's operand rows instead of x
int launch_mlp_fused_stat_split(acx_ctx* c, const BlockW& w, int C, const float* y, float* x, long long M, hipStream_t s, void* ln_out) {
    if (C != 96 || !w.wstream_h) ACX_FAIL(ACX_ERR_STATE, "stationary fused MLP: half streams were not packed for C=%d", C);
    ProfScope ps(c, ACX_K_MLP_FUSED, s);
    ACX_TRY(launch_stat_split<false>(w, 0, y, x, M, nullptr, s));
    return ln_out ? launch_stat_split<true>(w, 1, y, x, M, ln_out, s) : launch_stat_split<false>(w, 1, y, x, M, nullptr, s);
}

