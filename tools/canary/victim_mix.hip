// Victim with separable instruction classes: each class produces its own per-workgroup checksum.
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, unsigned long long* sums, int rounds, int n) {
    __shared__ float lds[4096];
    __shared__ float2 lds2[4096];
    __shared__ float4 lds4[2048];
    struct cf4 { float x, y; };                 // 4-byte aligned pair -> ds_write2_b32 / ds_read2_b32
    __shared__ cf4 ldsc[4096 + 512];
    const int tid = threadIdx.x;
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0, s8 = 0;
    unsigned long long s9 = 0;
    for (int r = 0; r < rounds; ++r) {
        const size_t base = ((size_t)blockIdx.x * rounds + r) * 4096 % (size_t)n;
        // class 0: scalar-width vector loads
        float a[4];
        for (int k = 0; k < 4; ++k) a[k] = in[base + tid + 256 * k];
        for (int k = 0; k < 4; ++k) s0 += __float_as_uint(a[k]);
        // class 1: 16-byte loads
        const float4 v = *reinterpret_cast<const float4*>(in + base + 4 * tid);
        s1 += __float_as_uint(v.x) + __float_as_uint(v.y) + __float_as_uint(v.z) + __float_as_uint(v.w);
        // class 7: 8-byte loads (complex samples, as FFT kernels read them), class 8: 8-byte stores + reload
        float2 c2[4];
        for (int k = 0; k < 4; ++k) c2[k] = *reinterpret_cast<const float2*>(in + base + 2 * (tid + 256 * k));
        for (int k = 0; k < 4; ++k) s7 += __float_as_uint(c2[k].x) + 3u * __float_as_uint(c2[k].y);
        // class 2: LDS exchange (write, barrier, strided read)
        for (int k = 0; k < 16; ++k) lds[tid + 256 * k] = a[k & 3] + (float)k;
        __syncthreads();
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += lds[(tid * 17 + k * 259) & 4095];
        __syncthreads();
        s2 += __float_as_uint(t);
        // class 5: 8-byte LDS exchange with a radix-8 style scatter (stride 9 float2), class 6: 16-byte LDS exchange
        for (int k = 0; k < 16; ++k) lds2[((tid + 256 * k) * 577) & 4095] = make_float2(a[k & 3] + k, a[(k + 1) & 3] - k);
        for (int k = 0; k < 8; ++k) lds4[((tid + 256 * k) * 263) & 2047] = make_float4(a[0] + k, a[1] + k, a[2] + k, a[3] + k);
        __syncthreads();
        float t2 = 0.f, t4 = 0.f;
        for (int k = 0; k < 16; ++k) { const float2 q = lds2[(tid + 256 * k) & 4095]; t2 += q.x - q.y; }
        for (int k = 0; k < 8; ++k) { const float4 q = lds4[(tid + 256 * k) & 2047]; t4 += q.x + q.y - q.z + q.w; }
        __syncthreads();
        s5 += __float_as_uint(t2); s6 += __float_as_uint(t4);
        // class 8: the FFT exchange shape: 4-byte-aligned pairs, padded index i + (i >> 3), 8 consecutive per lane out,
        //          stride-64 in (compiles to ds_write2_b32 / ds_read2_b32)
        for (int k = 0; k < 16; ++k) { const int i = ((tid * 16 + k) * 1) & 4095; cf4 q; q.x = a[k & 3] + k; q.y = a[(k + 2) & 3] - k; ldsc[i + (i >> 3)] = q; }
        __syncthreads();
        float tc = 0.f;
        for (int k = 0; k < 16; ++k) { const int i = (tid + 256 * k) & 4095; const cf4 q = ldsc[i + (i >> 3)]; tc += q.x * 1.5f - q.y; }
        __syncthreads();
        s9 += __float_as_uint(tc);
        // class 3: transcendental + fma chain
        float u = a[0];
        for (int k = 0; k < 8; ++k) u = __sinf(u) * 1.7f + __log2f(fabsf(u) + 1.5f);
        s3 += __float_as_uint(u);
        // class 4: cross-lane
        float w = a[1];
        for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o);
        s4 += __float_as_uint(w);
    }
    atomicAdd(&sums[0], s0 * (tid + 1)); atomicAdd(&sums[1], s1 * (tid + 1)); atomicAdd(&sums[2], s2 * (tid + 1));
    atomicAdd(&sums[3], s3 * (tid + 1)); atomicAdd(&sums[4], s4 * (tid + 1));
    atomicAdd(&sums[5], s5 * (tid + 1)); atomicAdd(&sums[6], s6 * (tid + 1)); atomicAdd(&sums[7], s7 * (tid + 1)); atomicAdd(&sums[8], s9 * (tid + 1));
}
extern "C" int victim_launch(const float* in, unsigned long long* sums, int blocks, int rounds, int n, void* stream) {
    victim<<<blocks, 256, 0, (hipStream_t)stream>>>(in, sums, rounds, n);
    return (int)hipGetLastError();
}
