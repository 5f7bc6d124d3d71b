// v_pk_fma_f32 rate against the register banks of its three 64-bit operands (lab, round 4).
//   hipcc -O3 --offload-arch=gfx950 tools/lab/pkfma_banks.hip -o build/pkfma_banks
// One wave per SIMD (256 threads per CU-filling workgroup) or two (512); 96 independent FMAs per loop iteration in a fixed
// register pattern: accumulators at v[ACC + 2i], sources a = v[A0 + 2 (i % 8)], b = v[B0 + 2 (i % 8)].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VAR>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
    // sources live in v200.., accumulators in v8..v199 (explicit registers: the banks are the experiment)
    asm volatile("v_mov_b32 v200, 1.0\n v_mov_b32 v201, 1.0\n v_mov_b32 v202, 0.5\n v_mov_b32 v203, 0.5\n v_mov_b32 v204, 0.25\n v_mov_b32 v205, 0.25\n v_mov_b32 v206, 0.125\n v_mov_b32 v207, 0.125\n"
                 "v_mov_b32 v208, 1.0\n v_mov_b32 v209, 1.0\n v_mov_b32 v210, 0.5\n v_mov_b32 v211, 0.5\n v_mov_b32 v212, 0.25\n v_mov_b32 v213, 0.25\n v_mov_b32 v214, 0.125\n v_mov_b32 v215, 0.125\n"
                 ::: "v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // 48 FMAs per asm block; VAR selects the register pattern
        if (VAR == 0) {        // acc%4==0, a%4==0, b%4==0   (all three pairs on banks 0,1)
            asm volatile(
#define F(acc, a, b) "v_pk_fma_f32 v[" #acc ":" #acc "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #acc ":" #acc "+1]\n"
                F(8,200,204) F(12,200,204) F(16,200,204) F(20,200,204) F(24,200,204) F(28,200,204) F(32,200,204) F(36,200,204)
                F(40,200,204) F(44,200,204) F(48,200,204) F(52,200,204) F(56,200,204) F(60,200,204) F(64,200,204) F(68,200,204)
                F(72,200,204) F(76,200,204) F(80,200,204) F(84,200,204) F(88,200,204) F(92,200,204) F(96,200,204) F(100,200,204)
                F(104,200,204) F(108,200,204) F(112,200,204) F(116,200,204) F(120,200,204) F(124,200,204) F(128,200,204) F(132,200,204)
                ::: "memory");
        } else if (VAR == 1) { // acc%4==0, a%4==2, b%4==0
            asm volatile(
                F(8,202,204) F(12,202,204) F(16,202,204) F(20,202,204) F(24,202,204) F(28,202,204) F(32,202,204) F(36,202,204)
                F(40,202,204) F(44,202,204) F(48,202,204) F(52,202,204) F(56,202,204) F(60,202,204) F(64,202,204) F(68,202,204)
                F(72,202,204) F(76,202,204) F(80,202,204) F(84,202,204) F(88,202,204) F(92,202,204) F(96,202,204) F(100,202,204)
                F(104,202,204) F(108,202,204) F(112,202,204) F(116,202,204) F(120,202,204) F(124,202,204) F(128,202,204) F(132,202,204)
                ::: "memory");
        } else if (VAR == 2) { // acc%4==0, a%4==2, b%4==2
            asm volatile(
                F(8,202,206) F(12,202,206) F(16,202,206) F(20,202,206) F(24,202,206) F(28,202,206) F(32,202,206) F(36,202,206)
                F(40,202,206) F(44,202,206) F(48,202,206) F(52,202,206) F(56,202,206) F(60,202,206) F(64,202,206) F(68,202,206)
                F(72,202,206) F(76,202,206) F(80,202,206) F(84,202,206) F(88,202,206) F(92,202,206) F(96,202,206) F(100,202,206)
                F(104,202,206) F(108,202,206) F(112,202,206) F(116,202,206) F(120,202,206) F(124,202,206) F(128,202,206) F(132,202,206)
                ::: "memory");
        } else if (VAR == 3) { // acc alternating 0 / 2 mod 4, a%4==0, b%4==2
            asm volatile(
                F(8,200,206) F(10,200,206) F(12,200,206) F(14,200,206) F(16,200,206) F(18,200,206) F(20,200,206) F(22,200,206)
                F(24,200,206) F(26,200,206) F(28,200,206) F(30,200,206) F(32,200,206) F(34,200,206) F(36,200,206) F(38,200,206)
                F(40,200,206) F(42,200,206) F(44,200,206) F(46,200,206) F(48,200,206) F(50,200,206) F(52,200,206) F(54,200,206)
                F(56,200,206) F(58,200,206) F(60,200,206) F(62,200,206) F(64,200,206) F(66,200,206) F(68,200,206) F(70,200,206)
                ::: "memory");
        } else if (VAR == 4) { // same source for a and b (one read port fewer?): acc%4==0, a = b, %4==2
            asm volatile(
                F(8,202,202) F(12,202,202) F(16,202,202) F(20,202,202) F(24,202,202) F(28,202,202) F(32,202,202) F(36,202,202)
                F(40,202,202) F(44,202,202) F(48,202,202) F(52,202,202) F(56,202,202) F(60,202,202) F(64,202,202) F(68,202,202)
                F(72,202,202) F(76,202,202) F(80,202,202) F(84,202,202) F(88,202,202) F(92,202,202) F(96,202,202) F(100,202,202)
                F(104,202,202) F(108,202,202) F(112,202,202) F(116,202,202) F(120,202,202) F(124,202,202) F(128,202,202) F(132,202,202)
                ::: "memory");
        } else {               // v_fma_f32 (unpacked), 32 of them, acc%4 cycling, for reference
            asm volatile(
#define G(acc, a, b) "v_fma_f32 v" #acc ", v" #a ", v" #b ", v" #acc "\n"
                G(8,200,205) G(9,200,205) G(10,200,205) G(11,200,205) G(12,200,205) G(13,200,205) G(14,200,205) G(15,200,205)
                G(16,200,205) G(17,200,205) G(18,200,205) G(19,200,205) G(20,200,205) G(21,200,205) G(22,200,205) G(23,200,205)
                G(24,200,205) G(25,200,205) G(26,200,205) G(27,200,205) G(28,200,205) G(29,200,205) G(30,200,205) G(31,200,205)
                G(32,200,205) G(33,200,205) G(34,200,205) G(35,200,205) G(36,200,205) G(37,200,205) G(38,200,205) G(39,200,205)
                ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("" ::: "v8","v9","v10","v11","v12","v16","v20","v24","v28","v32","v36","v40","v44","v48","v52","v56","v60","v64","v68","v72","v76","v80","v84","v88","v92","v96","v100","v104","v108","v112","v116","v120","v124","v128","v132","v133","v199");
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    if (out && threadIdx.x == 9999) out[0] = 1.f;
}

template <int VAR>
void run(const char* name, int threads) {
    unsigned long long* d; CK(hipMalloc(&d, 8));
    const int iters = 20000;
    k<VAR><<<256, threads>>>(nullptr, 100, d);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k<VAR><<<256, threads>>>(nullptr, iters, d);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost));
    const double n = 32.0 * iters;
    printf("%-44s %d waves/SIMD: %.2f cycles per instruction per wave, %.2f ns; SIMD: one per %.2f cycles\n", name, threads / 256, c / n, ms * 1e6 / n, c / n / (threads / 256));
    hipFree(d);
}

int main() {
    for (int t : {256, 512}) {
        run<0>("pk_fma acc 0, a 0, b 0 (mod 4)", t);
        run<1>("pk_fma acc 0, a 2, b 0", t);
        run<2>("pk_fma acc 0, a 2, b 2", t);
        run<3>("pk_fma acc 0/2 alternating, a 0, b 2", t);
        run<4>("pk_fma acc 0, a = b = 2", t);
        run<5>("v_fma_f32 (unpacked)", t);
    }
    return 0;
}
