// Stand-alone reproducer for the co-residency corruption of DESIGN.md 3b -- ONE file, no libacx, no torch.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/race2/standalone_repro.hip -o build/labs/standalone_repro
//   build/labs/standalone_repro            # prints wrong-thread counts of the victim per configuration
//
// aggressor (null stream)   a generic tiled fp16 GEMM loop, written here from scratch: every workgroup streams operand tiles
//                           into a 2-slot LDS ring by LDS-DMA (global_load_lds_dwordx4), waves read fragments with
//                           ds_read_b128 and run v_mfma_f32_32x32x16_f16 on LIVE data.  Two launch forms:
//                             shared     72 KB of LDS, < 128 registers: other workgroups fit beside it on a CU
//                             exclusive  160 KB of LDS claimed + 512 threads x 256 registers: nothing fits beside it
// victim (side stream)      register-only complex butterflies (radix-8 FFT stages iterated on registers, no LDS, no memory in
//                           the loop), compiled so that the arithmetic is PACKED FP32 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32)
//                           in one kernel and plain 32-bit VALU (inline asm keeps hipcc from packing) in the other.
// check                     the victim's result is compared, thread by thread, with the same kernel run ALONE (bitwise: the
//                           arithmetic is deterministic).  Any difference is a wrong result of a kernel that touches no
//                           memory the aggressor touches.
//
// What the library's guard (CU-exclusive workgroups, acx_internal.h) claims, and what this file lets anyone check without the
// library: "packed victim beside the SHARED aggressor: wrong threads > 0;  beside the EXCLUSIVE aggressor: 0;  scalar victim: 0".
// If the first count is 0 on a given box the corruption does not reproduce there with a generic aggressor -- then the product
// kernels of round 1 did something this file does not, and the erratum claim must be narrowed (the guard stays either way).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// ---------------------------------------------------------------- aggressor -------------------------------------------------
// C[256 x 128 per workgroup] += A[256 x K] . B[128 x K]^T, fp16 operands, 8 waves (4 x 2), K tiles of 32: per k-tile 16 KB (A) +
// 8 KB (B) arrive by LDS-DMA into slot (t & 1) while slot ((t + 1) & 1) is multiplied.  Results are written out (live data).
template <bool EXCLUSIVE>
__global__ __launch_bounds__(512) void aggressor(const _Float16* __restrict__ A, const _Float16* __restrict__ B, float* __restrict__ C, int K, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (EXCLUSIVE) asm volatile("" ::: "v255");          // 512 threads x 256 registers: the CU's whole register file
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;             // wave tile: rows 64 wm .. +63, columns 64 wn .. +63
    const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
    const _Float16* Ag = A + (size_t)bm * 256 * K;
    const _Float16* Bg = B + (size_t)bn * 128 * K;
    // slot layout: A tile [256 rows][32 k] fp16 = 64 B per row (16 KB), then B tile [128][32] (8 KB); 16-B pieces in lane order
    const int kSlot = 24 * 1024;
    auto stage = [&](int t, int slot) {
        // 24 pieces of 1 KB, three per wave: piece p covers 16 rows x 64 B; lane l -> row 16 p' + l / 4, 16-B chunk l % 4
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int p = wave + 8 * q;
            const bool isA = p < 16;
            const int pr = isA ? p : p - 16;
            const int row = 16 * pr + (lane >> 2), ch = lane & 3;
            const _Float16* src = (isA ? Ag : Bg) + (size_t)row * K + t * 32 + ch * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + slot * kSlot + p * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nt = K / 32;
    // three-slot ring (EXCLUSIVE: the same ring, the rest of the 160 KB is only claimed): tile t + 2 is requested while tile t
    // multiplies; a counted s_waitcnt vmcnt(3) + bare s_barrier per tile leaves the youngest tile in flight across the barrier
    stage(0, 0);
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nt; ++t) {
        if (t + 2 < nt) stage(t + 2, (t + 2) % 3);
        const char* a = smem + (t % 3) * kSlot;
        const char* b = a + 16 * 1024;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                 // two k-steps of 16
            h8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const h8*>(a + (64 * wm + 32 * i + (lane & 31)) * 64 + ks * 32 + (lane >> 5) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const h8*>(b + (64 * wn + 32 * j + (lane & 31)) * 64 + ks * 32 + (lane >> 5) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {            // three MFMAs per fragment pair, as the three partial products of the split arithmetic
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i ^ 1], bf[j], acc[i][j], 0, 0, 0);      // "lo x hi"
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j ^ 1], acc[i][j], 0, 0, 0);      // "hi x lo"
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);          // "hi x hi": one chain
                }
        }
        if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // GELU-like epilogue in PACKED fp32 (what the product GEMM's epilogue did in round 2): a few v_pk_mul / v_pk_fma + exp / rcp
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
        for (int r = 0; r < 16; r += 2) {
            f32x2 a; a.x = acc[i][j][r] * 1e-3f; a.y = acc[i][j][r + 1] * 1e-3f;
            f32x2 k1; k1.x = 0.23f; k1.y = 0.23f;
            f32x2 one; one.x = 1.f; one.y = 1.f;
            f32x2 den = a * a * k1 + one;
            f32x2 t; t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
            f32x2 e; e.x = __builtin_amdgcn_exp2f(-a.x * a.x); e.y = __builtin_amdgcn_exp2f(-a.y * a.y);
            f32x2 q = (t * k1 + one) * t * e;
            f32x2 g = a * q + a;
            acc[i][j][r] = g.x; acc[i][j][r + 1] = g.y;
        }
    float* Cg = C + ((size_t)bm * 256 + 64 * wm) * (128 * gridDim.x / tiles_m) + (size_t)bn * 128 + 64 * wn;
    const int ldc = 128 * (gridDim.x / tiles_m);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r)
        Cg[(size_t)(32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ldc + 32 * j + (lane & 31)] = acc[i][j][r];
}

// ---------------------------------------------------------------- victim ----------------------------------------------------
// Eight complex values per thread; per round: twiddle multiply, three radix-2 butterfly stages, rescale.  PACKED: complex
// numbers are f32x2 and the arithmetic is written with vector operators (hipcc emits v_pk_*_f32).  !PACKED: every operation
// goes through a one-instruction asm on scalars, which SLP cannot merge.
template <bool PACKED> struct Ops;
template <> struct Ops<true> {
    static __device__ __forceinline__ f32x2 mul(f32x2 a, f32x2 w) {           // (a.x + i a.y)(w.x + i w.y)
        f32x2 t; t.x = -a.y; t.y = a.x;
        f32x2 wx; wx.x = w.x; wx.y = w.x;
        f32x2 wy; wy.x = w.y; wy.y = w.y;
        return a * wx + t * wy;
    }
    static __device__ __forceinline__ f32x2 add(f32x2 a, f32x2 b) { return a + b; }
    static __device__ __forceinline__ f32x2 sub(f32x2 a, f32x2 b) { return a - b; }
    static __device__ __forceinline__ f32x2 scale(f32x2 a, float s) { f32x2 v; v.x = s; v.y = s; return a * v; }
};
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float s_mul(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <> struct Ops<false> {
    static __device__ __forceinline__ f32x2 mul(f32x2 a, f32x2 w) {
        f32x2 r;
        r.x = s_fma(-a.y, w.y, s_mul(a.x, w.x));
        r.y = s_fma(a.x, w.y, s_mul(a.y, w.x));
        return r;
    }
    static __device__ __forceinline__ f32x2 add(f32x2 a, f32x2 b) { f32x2 r; r.x = s_add(a.x, b.x); r.y = s_add(a.y, b.y); return r; }
    static __device__ __forceinline__ f32x2 sub(f32x2 a, f32x2 b) { f32x2 r; r.x = s_sub(a.x, b.x); r.y = s_sub(a.y, b.y); return r; }
    static __device__ __forceinline__ f32x2 scale(f32x2 a, float s) { f32x2 r; r.x = s_mul(a.x, s); r.y = s_mul(a.y, s); return r; }
};

template <bool PACKED>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ seed, float* __restrict__ out, int rounds) {
    using O = Ops<PACKED>;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    f32x2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { v[r].x = seed[(gid * 16 + 2 * r) & 0xffff]; v[r].y = seed[(gid * 16 + 2 * r + 1) & 0xffff]; }
    f32x2 w; w.x = 0.98078528f; w.y = -0.19509032f;
    f32x2 w8; w8.x = 0.70710678f; w8.y = -0.70710678f;
    f32x2 mi; mi.x = 0.f; mi.y = -1.f;
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = O::mul(v[r], w);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const f32x2 a = v[r], b = v[r + 4]; v[r] = O::add(a, b); v[r + 4] = O::sub(a, b); }
        v[5] = O::mul(v[5], w8); v[6] = O::mul(v[6], mi); v[7] = O::mul(O::mul(v[7], w8), mi);
#pragma unroll
        for (int g = 0; g < 8; g += 4)
#pragma unroll
            for (int r = 0; r < 2; ++r) { const f32x2 a = v[g + r], b = v[g + r + 2]; v[g + r] = O::add(a, b); v[g + r + 2] = O::sub(a, b); }
        v[3] = O::mul(v[3], mi); v[7] = O::mul(v[7], mi);
#pragma unroll
        for (int g = 0; g < 8; g += 2) { const f32x2 a = v[g], b = v[g + 1]; v[g] = O::add(a, b); v[g + 1] = O::sub(a, b); }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = O::scale(v[r], 0.35355339f);
    }
    float c = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) c = s_add(c, s_fma(3.f, v[r].y, v[r].x));       // (scalar in both builds: outside the loop)
    out[gid] = c;
}

// ---------------------------------------------------------------- driver ----------------------------------------------------
template <bool PACKED>
static int run_victim(const float* seed, float* out, int blocks, int rounds, hipStream_t st) {
    victim<PACKED><<<blocks, 256, 0, st>>>(seed, out, rounds);
    return (int)hipGetLastError();
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const int M = 256 * 64, N = 128 * 16, K = 4096;            // 1024 aggressor workgroups: four per CU
    const int vblocks = 1024, rounds = 20000;
    _Float16 *A, *B; float *C, *seed, *out, *ref;
    CHECK(hipMalloc(&A, (size_t)M * K * 2)); CHECK(hipMalloc(&B, (size_t)N * K * 2)); CHECK(hipMalloc(&C, (size_t)M * N * 4));
    CHECK(hipMalloc(&seed, 65536 * 4)); CHECK(hipMalloc(&out, vblocks * 256 * 4)); CHECK(hipMalloc(&ref, vblocks * 256 * 4));
    {
        std::vector<_Float16> h((size_t)M * K);
        unsigned s = 99u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 9 & 0x3ff) - 512) / 512.0f); }
        CHECK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
        std::vector<float> hs(65536);
        for (auto& v : hs) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9 & 0xffff) - 32768) / 32768.0f; }
        CHECK(hipMemcpy(seed, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    }
    CHECK(hipFuncSetAttribute((const void*)aggressor<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t side; CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    std::vector<float> h_ref(vblocks * 256), h_out(vblocks * 256);
    printf("victim: %d workgroups x 256 threads, %d rounds of register-only butterflies; aggressor: %d x %d x %d fp16 GEMM, %d workgroups\n",
           vblocks, rounds, M, N, K, (M / 256) * (N / 128));
    for (int excl = 0; excl <= 1; ++excl) {          // how dense is the aggressor?  (executed fp16 MFMA flops: 3 per tile pair)
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (excl) aggressor<true><<<(M / 256) * (N / 128), 512, 160 * 1024, 0>>>(A, B, C, K, M / 256);
            else aggressor<false><<<(M / 256) * (N / 128), 512, 72 * 1024, 0>>>(A, B, C, K, M / 256);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("aggressor alone, %s: %.3f ms = %.0f TFLOP/s of executed fp16 MFMA (%.2f of 2500)\n", excl ? "exclusive" : "shared", ms,
               3.0 * 2.0 * M * N * K / (ms * 1e-3) / 1e12, 3.0 * 2.0 * M * N * K / (ms * 1e-3) / 1e12 / 2500.0);
    }
    printf("%-10s %-12s %s\n", "victim", "aggressor", "wrong threads per repetition (of 262144)");
    for (int packed = 1; packed >= 0; --packed) {
        // the victim alone: the reference (and how long it lives)
        hipEvent_t v0, v1; CHECK(hipEventCreate(&v0)); CHECK(hipEventCreate(&v1));
        CHECK(hipEventRecord(v0, side));
        (packed ? run_victim<true> : run_victim<false>)(seed, ref, vblocks, rounds, side);
        CHECK(hipEventRecord(v1, side));
        CHECK(hipDeviceSynchronize());
        { float ms; CHECK(hipEventElapsedTime(&ms, v0, v1)); printf("%s victim alone: %.2f ms\n", packed ? "packed" : "scalar", ms); }
        CHECK(hipMemcpy(h_ref.data(), ref, h_ref.size() * 4, hipMemcpyDeviceToHost));
        for (int excl = -1; excl <= 1; ++excl) {        // -1: no aggressor (self-consistency), 0: shared, 1: exclusive
            printf("%-10s %-12s", packed ? "packed" : "scalar", excl < 0 ? "none" : (excl ? "exclusive" : "shared"));
            for (int rep = 0; rep < reps; ++rep) {
                CHECK(hipMemsetAsync(out, 0, h_out.size() * 4, side));
                CHECK(hipDeviceSynchronize());
                // the aggressor runs back to back for longer than the victim lives (~20 launches of 0.7 ms); the victim starts
                // after the first two launches are queued, so its workgroups find the CUs already occupied
                for (int a = 0; a < 24; ++a) {
                    if (excl == 0) aggressor<false><<<(M / 256) * (N / 128), 512, 72 * 1024, 0>>>(A, B, C, K, M / 256);
                    if (excl == 1) aggressor<true><<<(M / 256) * (N / 128), 512, 160 * 1024, 0>>>(A, B, C, K, M / 256);
                    if (a == 1) (packed ? run_victim<true> : run_victim<false>)(seed, out, vblocks, rounds, side);
                }
                if (excl < 0) (packed ? run_victim<true> : run_victim<false>)(seed, out, vblocks, rounds, side);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h_out.data(), out, h_out.size() * 4, hipMemcpyDeviceToHost));
                long wrong = 0;
                for (size_t i = 0; i < h_out.size(); ++i) wrong += std::memcmp(&h_out[i], &h_ref[i], 4) != 0;
                printf(" %6ld", wrong);
            }
            printf("\n");
        }
    }
    return 0;
}
