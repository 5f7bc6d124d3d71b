// Instrumented FFT victim for the split-GEMM <-> FFT interference (DESIGN.md "Open issue").
// The transform of frontend.hip (fft_core.h, one wave64 per frame, private LDS exchange buffers) with probes that
// tell the failure modes apart:
//   kind 1  own-write readback differs after the barrier (lost / overwritten write); the probe then re-reads the slot
//           up to 64 times: heal = number of extra reads until it matched (64 = never)
//   kind 2  two consecutive reads of the same exchange slot differ (transient read error)
//   kind 3  the twiddle table in LDS differs from its global copy when the workgroup ends (foreign write into a
//           read-only table)
//   kind 4  same as kind 1 for the second exchange (bufB), kind 5 same as kind 2 for bufB
//   kind 7  two consecutive reads of the same twiddle (read-only LDS table, heavily bank-conflicted access) differ
//   kind 8  the radix-8 butterfly computed twice from the same registers (inputs and twiddles laundered through an
//           empty asm so that the compiler cannot merge the two) differs: a transient VALU / register error
// Every event carries HW_ID / XCC_ID / LDS_ALLOC of the reporting wave.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <vector>
#include "../../audioset-convnext-inf_amd/csrc/fft_core.h"

using namespace acx;
constexpr int kWaves = 4;
constexpr int kEvWords = 16;
constexpr int kMaxEv = 256;

struct ProbeLds {
    float pad[1024];                 // stands where frontend.hip keeps the mel weights (same LDS footprint: 48 KB)
    cf tw[1024];
    cf buf[kWaves][2][kFftBufSlots];
};

__device__ __forceinline__ unsigned bits(float f) { return __float_as_uint(f); }

__device__ void report(unsigned* rep, unsigned kind, unsigned a, unsigned b, unsigned c, unsigned d, unsigned e,
                       unsigned f, unsigned g) {
    atomicAdd(&rep[kind], 1u);
    const unsigned slot = atomicAdd(&rep[15], 1u);
    if (slot >= kMaxEv) return;
    unsigned hwid, xcc, ldsa;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(ldsa));
    unsigned* ev = rep + 16 + slot * kEvWords;
    ev[0] = kind; ev[1] = blockIdx.x; ev[2] = threadIdx.x; ev[3] = a; ev[4] = b; ev[5] = c; ev[6] = d; ev[7] = e;
    ev[8] = f; ev[9] = g; ev[10] = hwid; ev[11] = xcc; ev[12] = ldsa;
}

// fft512_pass (fft_core.h) with the twiddles read twice and the butterfly computed twice
__device__ __forceinline__ int probed_pass(cf* v, int j, int Ns, const cf* tw1024, unsigned* rep, unsigned f) {
    const int k = j % Ns;
    const int step = k * (64 / Ns);
    cf tw[8], a[8], b[8];
    volatile const cf* vt = tw1024;
#pragma unroll
    for (int r = 1; r < 8; ++r) tw[r] = tw1024[2 * step * r];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 1; r < 8; ++r) {
        const float qx = vt[2 * step * r].x, qy = vt[2 * step * r].y;
        if (bits(qx) != bits(tw[r].x) || bits(qy) != bits(tw[r].y))
            report(rep, 7, f, (unsigned)(2 * step * r), bits(qx), bits(tw[r].x), bits(qy), bits(tw[r].y), (unsigned)Ns);
    }
    a[0] = v[0]; b[0] = v[0];
    asm volatile("" : "+v"(b[0].x), "+v"(b[0].y));
#pragma unroll
    for (int r = 1; r < 8; ++r) {
        a[r] = cf_mul(v[r], tw[r]);
        cf v2 = v[r], t2 = tw[r];
        asm volatile("" : "+v"(v2.x), "+v"(v2.y), "+v"(t2.x), "+v"(t2.y));
        b[r] = cf_mul(v2, t2);
    }
    fft8(a);
    __builtin_amdgcn_sched_barrier(0);
    fft8(b);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (bits(a[r].x) != bits(b[r].x) || bits(a[r].y) != bits(b[r].y))
            report(rep, 8, f, (unsigned)r, bits(a[r].x), bits(b[r].x), bits(a[r].y), bits(b[r].y), (unsigned)Ns);
        v[r] = a[r];
    }
    return (j / Ns) * Ns * 8 + k;
}

// Pure-register victim: the same butterflies iterated in registers, no LDS, no memory traffic inside the loop.
// out[gid] = checksum; identical launches must give identical checksums.
__global__ __launch_bounds__(256) void valu_victim_kernel(const float* __restrict__ seed, float* __restrict__ out, int rounds) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    cf v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = cf_make(seed[(gid * 16 + 2 * r) & 0xffff], seed[(gid * 16 + 2 * r + 1) & 0xffff]);
    const cf w = cf_make(0.98078528f, -0.19509032f);
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cf_mul(v[r], w);
        fft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r].x *= 0.35355339f; v[r].y *= 0.35355339f; }
    }
    float c = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) c += v[r].x + 3.f * v[r].y;
    out[gid] = c;
}
extern "C" int valu_victim_launch(const float* seed, float* out, int blocks, int rounds, void* stream) {
    valu_victim_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(seed, out, rounds);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void fft_probe_kernel(const float* __restrict__ wav, long long L, int T,
                                                        long long nframes, const float* __restrict__ hann,
                                                        const float* __restrict__ twiddle, float* __restrict__ power,
                                                        unsigned* __restrict__ rep) {
    __shared__ ProbeLds lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 256) { lds.tw[i] = cf_make(twiddle[2 * i], twiddle[2 * i + 1]); lds.pad[i] = (float)i; }
    float2 hw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) hw[r] = *reinterpret_cast<const float2*>(hann + 2 * (lane + 64 * r));
    __syncthreads();
    const long long per_iter = (long long)gridDim.x * kWaves;
    const long long iters = (nframes + per_iter - 1) / per_iter;
    for (long long it = 0; it < iters; ++it) {
        const long long f = it * per_iter + (long long)blockIdx.x * kWaves + wave;
        const bool valid = f < nframes;
        const long long b = valid ? f / T : 0;
        const int t = valid ? (int)(f - b * T) : 0;
        const float* x = wav + b * L;
        const long long p0 = 320LL * t;
        cf v[8], keep[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = 2 * (lane + 64 * r);
            v[r] = cf_make(x[reflect_index(p0 + n, L)] * hw[r].x, x[reflect_index(p0 + n + 1, L)] * hw[r].y);
        }
        cf* bufA = lds.buf[wave][0];
        cf* bufB = lds.buf[wave][1];
        volatile cf* vA = bufA;
        volatile cf* vB = bufB;
        int dst = fft512_pass(v, lane, 1, lds.tw);
#pragma unroll
        for (int r = 0; r < 8; ++r) { bufA[fft_pad(dst + r)] = v[r]; keep[r] = v[r]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = fft_pad(dst + r);
            float qx = vA[idx].x, qy = vA[idx].y;
            if (bits(qx) != bits(keep[r].x) || bits(qy) != bits(keep[r].y)) {
                unsigned heal = 0;
                for (; heal < 64; ++heal) {
                    __builtin_amdgcn_s_sleep(2);
                    const float rx = vA[idx].x, ry = vA[idx].y;
                    if (bits(rx) == bits(keep[r].x) && bits(ry) == bits(keep[r].y)) break;
                }
                report(rep, 1, (unsigned)f, (unsigned)idx, bits(qx), bits(keep[r].x), bits(qy), bits(keep[r].y), heal);
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = bufA[fft_pad(lane + 64 * r)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = fft_pad(lane + 64 * r);
            const float qx = vA[idx].x, qy = vA[idx].y;
            if (bits(qx) != bits(v[r].x) || bits(qy) != bits(v[r].y))
                report(rep, 2, (unsigned)f, (unsigned)idx, bits(qx), bits(v[r].x), bits(qy), bits(v[r].y), 0);
        }
        dst = probed_pass(v, lane, 8, lds.tw, rep, (unsigned)f);
#pragma unroll
        for (int r = 0; r < 8; ++r) { bufB[fft_pad(dst + r * 8)] = v[r]; keep[r] = v[r]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = fft_pad(dst + r * 8);
            float qx = vB[idx].x, qy = vB[idx].y;
            if (bits(qx) != bits(keep[r].x) || bits(qy) != bits(keep[r].y)) {
                unsigned heal = 0;
                for (; heal < 64; ++heal) {
                    __builtin_amdgcn_s_sleep(2);
                    const float rx = vB[idx].x, ry = vB[idx].y;
                    if (bits(rx) == bits(keep[r].x) && bits(ry) == bits(keep[r].y)) break;
                }
                report(rep, 4, (unsigned)f, (unsigned)idx, bits(qx), bits(keep[r].x), bits(qy), bits(keep[r].y), heal);
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = bufB[fft_pad(lane + 64 * r)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = fft_pad(lane + 64 * r);
            const float qx = vB[idx].x, qy = vB[idx].y;
            if (bits(qx) != bits(v[r].x) || bits(qy) != bits(v[r].y))
                report(rep, 5, (unsigned)f, (unsigned)idx, bits(qx), bits(v[r].x), bits(qy), bits(v[r].y), 0);
        }
        dst = probed_pass(v, lane, 64, lds.tw, rep, (unsigned)f);
#pragma unroll
        for (int r = 0; r < 8; ++r) bufA[fft_pad(dst + r * 64)] = v[r];
        __syncthreads();
        float* P = reinterpret_cast<float*>(bufB);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = lane + 64 * r;
            const cf X = rfft1024_bin(bufA, k, lds.tw);
            P[k] = X.x * X.x + X.y * X.y;
        }
        __syncthreads();
        if (valid)
#pragma unroll
            for (int r = 0; r < 8; ++r) power[f * 512 + lane + 64 * r] = P[lane + 64 * r];
        __syncthreads();
    }
    for (int i = tid; i < 1024; i += 256) {
        const cf q = lds.tw[i];
        if (bits(q.x) != bits(twiddle[2 * i]) || bits(q.y) != bits(twiddle[2 * i + 1]))
            report(rep, 3, 0, (unsigned)i, bits(q.x), bits(twiddle[2 * i]), bits(q.y), bits(twiddle[2 * i + 1]), 0);
        if (bits(lds.pad[i]) != bits((float)i)) report(rep, 6, 0, (unsigned)i, bits(lds.pad[i]), bits((float)i), 0, 0, 0);
    }
}

static float *g_hann = nullptr, *g_tw = nullptr;
extern "C" int probe_init() {
    std::vector<float> hann(1024), tw(2048);
    for (int n = 0; n < 1024; ++n) {
        hann[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / 1024));
        tw[2 * n] = (float)std::cos(-2.0 * M_PI * n / 1024); tw[2 * n + 1] = (float)std::sin(-2.0 * M_PI * n / 1024);
    }
    if (hipMalloc(&g_hann, 4096) != hipSuccess || hipMalloc(&g_tw, 8192) != hipSuccess) return -1;
    hipMemcpy(g_hann, hann.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(g_tw, tw.data(), 8192, hipMemcpyHostToDevice);
    return 0;
}
// power: (nframes, 512) floats; rep: 16 + 256*16 unsigned words (zeroed by the caller)
extern "C" int probe_launch(const float* wav, long long L, int B, float* power, unsigned* rep, void* stream) {
    const int T = (int)(L / 320 + 1);
    const long long nframes = (long long)B * T;
    long long blocks = (nframes + kWaves - 1) / kWaves;
    if (blocks > 4096) blocks = 4096;
    fft_probe_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(wav, L, T, nframes, g_hann, g_tw, power, rep);
    return (int)hipGetLastError();
}
