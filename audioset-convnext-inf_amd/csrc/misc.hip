// K6 -- tail of forward_features + head (convnext.py:279-285, :321-325), and the NHWC->NCHW
// transpose that gives forward_frame_embeddings its layout (convnext.py:276-277).
#include "acx_internal.h"

namespace acx {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup of 768 threads per clip.  x NHWC (B,H3,7,768):
//   mean over the 7 frequency columns (torch.mean(x, dim=3)), then max over time + mean over time,
//   nn.LayerNorm(768, eps=1e-6) -> scene embedding; Linear 768->527 -> logits; sigmoid -> probs.
// Thread = (float4 of channels, one of 4 time phases): every thread streams ~H3/4 rows x 7 columns of
// independent 16-B loads (the first version walked all H3 rows serially per thread: latency-bound, 143 us),
// partial (max, sum) meet in LDS.
__global__ __launch_bounds__(768) void pool_head_kernel(const float* __restrict__ x, int H3,
                                                        const float* __restrict__ nw, const float* __restrict__ nb,
                                                        const float* __restrict__ hw, const float* __restrict__ hb,
                                                        float* __restrict__ scene, float* __restrict__ logits,
                                                        float* __restrict__ probs) {
    __shared__ __attribute__((aligned(16))) float pmax[4][768];
    __shared__ __attribute__((aligned(16))) float psum[4][768];
    __shared__ __attribute__((aligned(16))) float emb[768];
    __shared__ float red[12];
    const int tid = threadIdx.x;
    const int cg = tid % 192, ph = tid / 192;
    const long long b = blockIdx.x;
    const float* xb = x + b * (long long)H3 * 7 * 768 + 4 * cg;
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), sm = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = ph; h < H3; h += 4) {
        const float* r = xb + (long long)h * 7 * 768;
        float4 v[7];
#pragma unroll
        for (int w = 0; w < 7; ++w) v[w] = *reinterpret_cast<const float4*>(r + w * 768);
        float4 s = v[0];
#pragma unroll
        for (int w = 1; w < 7; ++w) { s.x += v[w].x; s.y += v[w].y; s.z += v[w].z; s.w += v[w].w; }
        s.x *= (1.0f / 7.0f); s.y *= (1.0f / 7.0f); s.z *= (1.0f / 7.0f); s.w *= (1.0f / 7.0f);
        mx.x = fmaxf(mx.x, s.x); mx.y = fmaxf(mx.y, s.y); mx.z = fmaxf(mx.z, s.z); mx.w = fmaxf(mx.w, s.w);
        sm.x += s.x; sm.y += s.y; sm.z += s.z; sm.w += s.w;
    }
    *reinterpret_cast<float4*>(&pmax[ph][4 * cg]) = mx;
    *reinterpret_cast<float4*>(&psum[ph][4 * cg]) = sm;
    __syncthreads();
    // thread c: combine the 4 phases (time order of the sum is (h%4, h/4) -- fp32 re-association only)
    const int c = tid;
    const float m = fmaxf(fmaxf(pmax[0][c], pmax[1][c]), fmaxf(pmax[2][c], pmax[3][c]));
    const float su = (psum[0][c] + psum[1][c]) + (psum[2][c] + psum[3][c]);
    const float pooled = m + su / (float)H3;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) t += red[i];
        return t;
    };
    const float mean = block_sum(pooled) * (1.0f / 768.0f);
    const float d = pooled - mean;
    const float var = block_sum(d * d) * (1.0f / 768.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    const float e0 = fmaf(d * rstd, nw[c], nb[c]);
    emb[c] = e0;
    if (scene) scene[b * 768 + c] = e0;
    __syncthreads();
    if (!logits && !probs) return;
    const int lane = tid & 63, wave = tid >> 6;
    float4 e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) e[k] = *reinterpret_cast<const float4*>(&emb[4 * (lane + 64 * k)]);
    // four rows of the head per wave and pass: 12 independent 16-byte loads per lane in flight (one row at a time was a chain
    // of 44 L2 round trips per wave)
    for (int n0 = wave; n0 < kClasses; n0 += 48) {
        float4 w4[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 12 * r < kClasses ? n0 + 12 * r : n0;
            const float4* wr = reinterpret_cast<const float4*>(hw + (long long)n * 768);
#pragma unroll
            for (int k = 0; k < 3; ++k) w4[r][k] = wr[lane + 64 * k];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s = fmaf(e[k].x, w4[r][k].x, s); s = fmaf(e[k].y, w4[r][k].y, s);
                s = fmaf(e[k].z, w4[r][k].z, s); s = fmaf(e[k].w, w4[r][k].w, s);
            }
            s = wave_sum(s);
            const int n = n0 + 12 * r;
            if (lane == 0 && n < kClasses) {
                const float z = s + hb[n];
                if (logits) logits[b * kClasses + n] = z;
                if (probs) probs[b * kClasses + n] = 1.0f / (1.0f + expf(-z));
            }
        }
    }
}

int launch_pool_head(acx_ctx* c, const float* x, int B, int H3, float* scene, float* logits, float* probs,
                     hipStream_t s) {
    ProfScope ps(c, ACX_K_POOLHEAD, s);
    launch_kernel(&pool_head_kernel, dim3(B), dim3(768), 0, s, x, H3, c->d_norm_w, c->d_norm_b, c->d_head_w, c->d_head_b,
                                                   scene, logits, probs);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// out[b][c][p] = x[b][p][c], p = h*W + w.  32x32 tiles through LDS (+1 pad), coalesced both ways.
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int P, int C) {
    __shared__ float t[32][33];
    const long long b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const float* xb = x + b * (long long)P * C;
    float* ob = out + b * (long long)P * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + ty + 8 * i;
        if (p < P) t[ty + 8 * i][tx] = xb[(long long)p * C + c0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i;
        const int p = p0 + tx;
        if (p < P) ob[(long long)c * P + p] = t[tx][ty + 8 * i];
    }
}

int launch_nhwc_to_nchw(acx_ctx* c, const float* x, float* out, int B, int H, int W, int C, hipStream_t s) {
    if (C % 32 != 0) ACX_FAIL(ACX_ERR_SHAPE, "nhwc_to_nchw: C=%d is not a multiple of 32", C);
    const int P = H * W;
    ProfScope ps(c, ACX_K_TRANSPOSE, s);
    launch_kernel(&nhwc_to_nchw_kernel, dim3((P + 31) / 32, C / 32, B), dim3(256), 0, s, x, out, P, C);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// Stored clips -> model input: AudioSet shards hold int16 PCM and the reference scales by 1 / 32767 on the host
// (utilities.py:226-227 `int16_to_float32`: (x / 32767.0).astype(np.float32), numpy promotes to float64).  Here the clips cross
// PCIe as int16 and are widened on the GPU: float(double(x) / 32767.0) -- the same two roundings, so the same bits -- 8 samples
// (16 bytes in, 32 out) per thread and step; the tail of an odd-length buffer sample by sample.
__global__ __launch_bounds__(256) void pcm16_to_f32_kernel(const short* __restrict__ in, float* __restrict__ out, long long n) {
    const long long n8 = n >> 3;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const uint4 u = reinterpret_cast<const uint4*>(in)[i];
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
        float r[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            r[2 * q] = (float)((double)(short)(w[q] & 0xffffu) / 32767.0);
            r[2 * q + 1] = (float)((double)(short)(w[q] >> 16) / 32767.0);
        }
        reinterpret_cast<float4*>(out)[2 * i] = make_float4(r[0], r[1], r[2], r[3]);
        reinterpret_cast<float4*>(out)[2 * i + 1] = make_float4(r[4], r[5], r[6], r[7]);
    }
    for (long long i = (n8 << 3) + blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (float)((double)in[i] / 32767.0);
}

int launch_pcm16_to_f32(const short* in, float* out, long long n, hipStream_t s) {
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        ACX_FAIL(ACX_ERR_ARG, "acx_pcm16_to_f32: buffers must be 16-byte aligned");
    long long blocks = (n / 8 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    launch_kernel(&pcm16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

// element-wise fp32 <-> bf16 (round to nearest even): the per-layer entry points of the C ABI keep fp32 tensors in every mode
// and convert at their boundary when the activations live in HBM as bf16 (ACX_PREC_BF16_ACT)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        reinterpret_cast<uint2*>(out)[i] = uint2{acx_pack_bf16x2(v.x, v.y), acx_pack_bf16x2(v.z, v.w)};
    }
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const __bf16* __restrict__ in, float* __restrict__ out, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const uint2 u = reinterpret_cast<const uint2*>(in)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(acx_bf16_lo(u.x), acx_bf16_hi(u.x), acx_bf16_lo(u.y), acx_bf16_hi(u.y));
    }
}
int launch_convert_f32_to_bf16(const float* in, void* out, long long n, hipStream_t s) {
    if (n % 4 != 0) ACX_FAIL(ACX_ERR_SHAPE, "convert: %lld elements (not a multiple of 4)", n);
    const long long n4 = n / 4;
    if (n4 == 0) return ACX_OK;
    const long long blocks = (n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096;
    launch_kernel(&f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, reinterpret_cast<__bf16*>(out), n4);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}
int launch_convert_bf16_to_f32(const void* in, float* out, long long n, hipStream_t s) {
    if (n % 4 != 0) ACX_FAIL(ACX_ERR_SHAPE, "convert: %lld elements (not a multiple of 4)", n);
    const long long n4 = n / 4;
    if (n4 == 0) return ACX_OK;
    const long long blocks = (n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096;
    launch_kernel(&bf16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const __bf16*>(in), out, n4);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

}  // namespace acx
