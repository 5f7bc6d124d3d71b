// Lab: the 16-block 4x4x4 bf16 MFMA (one 4x4x4 product per group of four lanes) as the engine of a depthwise conv --
// the register layout, checked against a host sum, and its issue rate alone, beside v_perm_b32 and beside ds_read_b64, at
// 1 / 2 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/lab/mfma4_rates.hip -o /tmp/mfma4_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

static uint16_t to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16); }
static float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void layout_kernel(const s4* a, const s4* b, f4* d) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}

// MODE 0: MFMAs alone (8 independent accumulators); 1: one v_perm_b32 per MFMA; 2: two per MFMA; 3: one ds_read_b64 per MFMA;
// 4: one perm and half a ds_read_b64 per MFMA
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned s) {
    __shared__ unsigned long long lds[2048];
    lds[threadIdx.x] = threadIdx.x * 0x0001000100010001ull; lds[threadIdx.x + 256] = s;
    __syncthreads();
    f4 acc[8];
    s4 a[8], b[8];
    unsigned p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        a[i] = s4{(short)(0x3f80 + i), (short)(0x3f00 + threadIdx.x), (short)0x3e80, (short)(0x3f80 + s)};
        b[i] = s4{(short)(0x3f00 + i), (short)(0x3f80 + s), (short)0x3f00, (short)(0x3e80 + threadIdx.x)};
        p[i] = threadIdx.x * 77u + i;
    }
    unsigned long long ld = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[u], b[(u + 3) & 7], acc[u], 0, 0, 0);
            if (MODE == 1 || MODE == 2 || MODE == 4) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(p[u]) : "v"(p[(u + 1) & 7]), "v"(p[(u + 5) & 7]));
            if (MODE == 2) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(p[(u + 2) & 7]) : "v"(p[(u + 3) & 7]), "v"(p[(u + 6) & 7]));
            if (MODE == 3 || (MODE == 4 && (u & 1))) {
                unsigned long long v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)((threadIdx.x * 8 + u * 2048 + it * 8) & 16383)));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                ld ^= v;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float t = (float)(unsigned)ld;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (float)p[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// MFMAs alone, pinned by inline asm (the builtin form of MODE 0 lets hipcc rotate accumulators through v_accvgpr moves): NACC
// accumulators used round-robin -- the distance between two instructions on the same accumulator
template <int NACC>
__global__ __launch_bounds__(256) void chain_kernel(float* out, int iters, unsigned s) {
    f4 acc[NACC];
    s4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = s4{(short)(0x3f80 + i), (short)(0x3f00 + threadIdx.x), (short)0x3e80, (short)(0x3f80 + s)};
        b[i] = s4{(short)(0x3f00 + i), (short)(0x3f80 + s), (short)0x3f00, (short)(0x3e80 + threadIdx.x)};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+a"(acc[u % NACC]) : "v"(a[u & 3]), "v"(b[(u + 1) & 3]));
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
    // ---- layout
    std::vector<uint16_t> ha(64 * 4), hb(64 * 4);
    for (int i = 0; i < 256; ++i) { ha[i] = to_bf16((float)((i * 7) % 13 - 6)); hb[i] = to_bf16((float)((i * 5) % 11 - 5)); }
    s4 *da, *db; f4* dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(da, db, dd);
    std::vector<float> hd(256);
    hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost);
    // guess: lane = 4 * block + r;  A lane holds row i = r, k = 0..3;  B lane holds column j = r, k = 0..3;  D lane holds column j = r, register = row i
    int bad = 0;
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float ref = 0.f;
                for (int k = 0; k < 4; ++k) ref += from_bf16(ha[(4 * blk + i) * 4 + k]) * from_bf16(hb[(4 * blk + j) * 4 + k]);
                if (hd[(4 * blk + j) * 4 + i] != ref) ++bad;
            }
    printf("layout guess D[lane 4 blk + j][reg i] = sum_k A[lane 4 blk + i][k] B[lane 4 blk + j][k]: %s (%d of 256 wrong)\n", bad ? "WRONG" : "right", bad);
    if (bad) {
        int bad2 = 0;       // the other guess: blocks interleaved, lane = 16 r + block
        for (int blk = 0; blk < 16; ++blk)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    float ref = 0.f;
                    for (int k = 0; k < 4; ++k) ref += from_bf16(ha[(16 * i + blk) * 4 + k]) * from_bf16(hb[(16 * j + blk) * 4 + k]);
                    if (hd[(16 * j + blk) * 4 + i] != ref) ++bad2;
                }
        printf("layout guess lane = 16 r + block: %s (%d wrong)\n", bad2 ? "WRONG" : "right", bad2);
    }
    // ---- rates
    float* out; hipMalloc(&out, 256 * 2 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[5] = {"mfma 4x4x4 bf16 alone", "+ 1 v_perm_b32 per MFMA", "+ 2 v_perm_b32 per MFMA", "+ 1 ds_read_b64 per MFMA", "+ 1 perm + 1/2 ds_read_b64"};
    for (int mode = 0; mode < 5; ++mode)
        for (int wps : {1, 2}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                dim3 g(256 * wps), b(256);
                if (mode == 0) rate_kernel<0><<<g, b>>>(out, iters, 3u);
                else if (mode == 1) rate_kernel<1><<<g, b>>>(out, iters, 3u);
                else if (mode == 2) rate_kernel<2><<<g, b>>>(out, iters, 3u);
                else if (mode == 3) rate_kernel<3><<<g, b>>>(out, iters, 3u);
                else rate_kernel<4><<<g, b>>>(out, iters, 3u);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
            }
            const double ns = best * 1e6 / ((double)iters * 8 * wps);
            printf("%-30s waves/SIMD=%d  %8.3f ms  %6.2f ns per MFMA and SIMD  = %5.1f GMAC/s per SIMD (1024 MACs each), chip %6.1f TMAC/s\n",
                   names[mode], wps, best, ns, 1024.0 / ns, 1024.0 / ns * 1024 / 1000);
        }
    for (int nacc : {1, 2, 4, 8, 16})
        for (int wps : {1, 2}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                dim3 g(256 * wps), b(256);
                if (nacc == 1) chain_kernel<1><<<g, b>>>(out, iters, 3u);
                else if (nacc == 2) chain_kernel<2><<<g, b>>>(out, iters, 3u);
                else if (nacc == 4) chain_kernel<4><<<g, b>>>(out, iters, 3u);
                else if (nacc == 8) chain_kernel<8><<<g, b>>>(out, iters, 3u);
                else chain_kernel<16><<<g, b>>>(out, iters, 3u);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
            }
            const double ns = best * 1e6 / ((double)iters * 16 * wps);
            printf("mfma 4x4x4 bf16 pinned, %2d accumulators round-robin  waves/SIMD=%d  %8.3f ms  %6.2f ns per MFMA and SIMD\n", nacc, wps, best, ns);
        }
    return 0;
}
