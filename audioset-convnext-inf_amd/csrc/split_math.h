// Shared by the split-fp16 and bf16 kernels: vector types, LDS-DMA helpers, the GELU and the fp32 -> (fp16 hi, fp16 lo) split.
#pragma once
#include <hip/hip_runtime.h>

namespace acx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// One 1-KB LDS-DMA piece (global_load_lds_dwordx4: lane l's 16 bytes from gsrc land at lds_dst + 16 l) issued from inline
// asm, so that hipcc does NOT know an LDS write is in flight: next to the builtin form it orders the next LDS read of the
// wave behind the DMA with s_waitcnt vmcnt(0) -- which also waits for every store and load the wave has in flight (a full
// HBM round trip at each tile boundary of a persistent kernel).  The caller owns the ordering: a counted s_waitcnt vmcnt(N)
// of its own (+ a barrier for other waves) between the piece and the reads of its bytes.  M0 carries the LDS base and is
// restored (cdna_hip_programming.md 5.7).  lds_dst must be wave-uniform.
__device__ __forceinline__ void acx_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The same without saving M0 (three instructions instead of five): only for kernels in which nothing else uses M0 -- no
// LDS-DMA builtin, no s_movrel, no GWS; the build checks the kernel's ISA for M0 uses outside these statements
// (tools/check_exclusive.py).
__device__ __forceinline__ void acx_glds16_own_m0(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
}
// The same with a SCALAR base and a 32-bit lane offset (global_load_lds_dwordx4 voff, s[base:base+1]): for a piece whose source is
// contiguous -- lane l fetches base + voff(l) -- the 64-bit address per lane of the form above is not needed, and the issue of the
// piece is several times cheaper: the paired bf16 MLP spends 15 % of its time on the per-lane-address form and none measurable on
// this one (round 5, profiles/r05_e_lds_dma_saddr.txt).  Same M0 contract as acx_glds16_own_m0; gbase must be wave-uniform.
__device__ __forceinline__ void acx_glds16_s(const void* gbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
// a wave-uniform pointer that came out of vector arithmetic (64-bit products are vector instructions even on uniform values)
// -> scalar registers, for the "s" operand above
__device__ __forceinline__ const char* acx_scalar_ptr(const void* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
// A RUN of pieces that are 1 KB apart in the stream AND in the LDS (a wave's pieces of one segment): the immediate offset of
// global_load_lds applies to both addresses (verified on the part: offset:1024 moves source and destination by 1 KB), so one
// M0 write + one 64-bit base serve up to eight pieces (13-bit signed immediate: -4096 .. 3072) and a piece is ONE instruction
// instead of five (v_lshl_add_u64, s_add, s_mov m0, s_nop, the load) -- in a loop where every issued instruction of the single
// wave per SIMD costs ~3-4 exposed cycles (profiles/r03_r_wide384_cycles.txt).  Same M0 contract as acx_glds16_own_m0.
__device__ __forceinline__ void acx_set_m0(unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds_dst) : "memory");
}
// piece j (0..7) of the run whose base points at piece 4: offset (j - 4) KB.  j must fold to a constant (unrolled loops): the
// immediate is part of the instruction.
__device__ __forceinline__ void acx_glds16_run(const void* gbase, int j) {
#define ACX_GLDS_CASE(J_) case J_: asm volatile("global_load_lds_dwordx4 %0, off offset:%1" :: "v"(gbase), "n"(((J_) - 4) * 1024) : "memory"); break;
    switch (j) {
        ACX_GLDS_CASE(0) ACX_GLDS_CASE(1) ACX_GLDS_CASE(2) ACX_GLDS_CASE(3)
        ACX_GLDS_CASE(4) ACX_GLDS_CASE(5) ACX_GLDS_CASE(6) ACX_GLDS_CASE(7)
        default: __builtin_unreachable();
    }
#undef ACX_GLDS_CASE
}
// The run form with a scalar base (see acx_glds16_s): piece j (0..7) of the run whose scalar base points at piece 4
__device__ __forceinline__ void acx_glds16_run_s(const void* sbase, unsigned voff, int j) {
#define ACX_GLDS_CASE(J_) case J_: asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" :: "v"(voff), "s"(sbase), "n"(((J_) - 4) * 1024) : "memory"); break;
    switch (j) {
        ACX_GLDS_CASE(0) ACX_GLDS_CASE(1) ACX_GLDS_CASE(2) ACX_GLDS_CASE(3)
        ACX_GLDS_CASE(4) ACX_GLDS_CASE(5) ACX_GLDS_CASE(6) ACX_GLDS_CASE(7)
        default: __builtin_unreachable();
    }
#undef ACX_GLDS_CASE
}
// on ? a : b without a select the compiler could turn into a branch (a branch inside a hand-placed loop body splits it into
// basic blocks and lets code sink out of its MFMA gaps)
__device__ __forceinline__ unsigned acx_pick(bool on, unsigned a, unsigned b) {
    const unsigned m = 0u - (unsigned)on;
    return b ^ ((a ^ b) & m);
}
__device__ __forceinline__ unsigned acx_lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(p);
}

// fp16(g - float(half of h)) written straight into one half of a packed register: v_fma_mixlo_f16 / v_fma_mixhi_f16 (the fma is
// evaluated in fp32 -- exactly, the operands are within a factor 2^11 -- and rounded once, as v_fma_mix_f32 + v_cvt_pk_f16_f32
// did with one instruction more per pair; bit-identical on 2^20 random pairs including the subnormal range)
__device__ __forceinline__ void acx_lo_half_lower(unsigned& lo, float gx, unsigned h) {
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(gx), "v"(h));
}
__device__ __forceinline__ void acx_lo_half_upper(unsigned& lo, float gy, unsigned h) {
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(gy), "v"(h));
}
// (x, y) -> packed fp16 hi halves and packed fp16 lo halves (lo = fp16(v - float(hi))): 3 instructions for two values
__device__ __forceinline__ void acx_split_pair(const float x, const float y, unsigned& hi, unsigned& lo) {
    f32x2 v; v.x = x; v.y = y;
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
    acx_lo_half_lower(lo, x, hi);
    acx_lo_half_upper(lo, y, hi);
}

// ---- GELU + split (third form, round 3; rounds 1-2: A&S 7.1.26 in 17-19 packed, then 15 scalar instructions per element -- git
// history): 10.5 vector instructions per element, ONE transcendental --------------------------
//   erfc(|v| / sqrt 2) ~= E(|v|) = exp2(-|v| Q(|v|)),  Q of degree 4 (five coefficients, leading one positive: -|v| Q -> -inf
//   for large |v|, E -> 0 without a clamp; E(0) = 1 exactly, so the relative accuracy near 0 is kept)
//   gelu(v) = 0.5 v + 0.5 |v| (1 - E):  minimax fit of 0.5 |v| (E - erfc) over [0, 9] (tools/lab/fit_gelu.py):
//   |gelu error| <= 5.4e-7 in exact arithmetic, <= 1e-6 as evaluated in fp32 (the fp32 rounding of a result near 4 is
//   4.8e-7), against 2.1e-7 for A&S 7.1.26 (the native-fp32 path, gemm.hip) -- whose 1 / (1 + p |v|) and exp(-v^2 / 2) cost a second transcendental
//   (8 issue cycles each against 2-4 for an FMA, profiles/r03_c_valu_opcode_costs.txt) and four more instructions.
// Unit: z = a (sinv 0.5 kH) = 0.5 kH v, the linear term of the result itself (one exact power-of-two multiply):
//   g = gelu(v) kH = z + |z| (1 - E),   exp2 argument = |z| (K0 + K1 |z| + ... + K4 |z|^4),  Kj = -cj / (0.5 kH)^(j+1)
// (exact scalings of the five constants; 0.5 kH = 2^-25 .. 2^11 keeps every Kj a normal number, api.hip hidden_scale_for).
// For v << 0 the two terms cancel with an absolute error of |v| kH 2^-25 -- what the reference's own fp32 evaluation of
// 0.5 v (1 + erf) carries there.  The fp32 -> fp16 hi / lo split is v_cvt_pk_f16_f32 for the hi halves and one v_fma_mixlo_f16 /
// v_fma_mixhi_f16 per element for fp16(g - float(hi)).  TWENTY-ONE single-instruction steps per register pair (two values),
// four registers of state.
struct GeluK3 { float zs, k0, k1, k2, k3, k4; };
__device__ __forceinline__ GeluK3 gelu_k3(float sinv, float kh) {
    GeluK3 k;
    const float h = 0.5f * kh, ih = 1.0f / h;       // powers of two
    k.zs = sinv * h;
    float s = ih;
    k.k0 = -1.1510010957717896f * s; s *= ih;
    k.k1 = -0.4595935642719269f * s; s *= ih;
    k.k2 = -0.05214935168623924f * s; s *= ih;
    k.k3 = 0.00719997426494956f * s; s *= ih;
    k.k4 = -0.0004882981302216649f * s;
    return k;
}
// Two waves per SIMD (mlp_fused_split.hip): a VALU instruction with a scalar-register operand issues at 4.2 cycles, the same
// instruction on vector registers only at 2.2 (profiles/r03_c_valu_opcode_costs.txt) -- pin the constants in VGPRs there.  At one
// wave per SIMD both forms cost the same; the wide kernels keep them scalar.
__device__ __forceinline__ void gelu_k3_to_vgprs(GeluK3& k) {
    asm volatile("" : "+v"(k.zs), "+v"(k.k0), "+v"(k.k1), "+v"(k.k2), "+v"(k.k3), "+v"(k.k4));
}
struct GeluState3 { float zx, zy, qx, qy; };
constexpr int kGelu3Nano = 21;      // steps per register pair
// step I of the pair (ax, ay) -> packed halves (hi, lo); ax / ay are read by steps 0 and 1 only
template <int I>
__device__ __forceinline__ void gelu3_nano(GeluState3& s, const GeluK3 k, const float ax, const float ay, unsigned& hi, unsigned& lo) {
    if constexpr (I == 0) s.zx = ax * k.zs;
    else if constexpr (I == 1) s.zy = ay * k.zs;
    else if constexpr (I == 2) s.qx = __builtin_fmaf(__builtin_fabsf(s.zx), k.k4, k.k3);
    else if constexpr (I == 3) s.qy = __builtin_fmaf(__builtin_fabsf(s.zy), k.k4, k.k3);
    else if constexpr (I == 4) s.qx = __builtin_fmaf(s.qx, __builtin_fabsf(s.zx), k.k2);
    else if constexpr (I == 5) s.qy = __builtin_fmaf(s.qy, __builtin_fabsf(s.zy), k.k2);
    else if constexpr (I == 6) s.qx = __builtin_fmaf(s.qx, __builtin_fabsf(s.zx), k.k1);
    else if constexpr (I == 7) s.qy = __builtin_fmaf(s.qy, __builtin_fabsf(s.zy), k.k1);
    else if constexpr (I == 8) s.qx = __builtin_fmaf(s.qx, __builtin_fabsf(s.zx), k.k0);
    else if constexpr (I == 9) s.qy = __builtin_fmaf(s.qy, __builtin_fabsf(s.zy), k.k0);
    else if constexpr (I == 10) s.qx *= __builtin_fabsf(s.zx);
    else if constexpr (I == 11) s.qy *= __builtin_fabsf(s.zy);
    else if constexpr (I == 12) s.qx = __builtin_amdgcn_exp2f(s.qx);
    else if constexpr (I == 13) s.qy = __builtin_amdgcn_exp2f(s.qy);
    else if constexpr (I == 14) s.qx = 1.0f - s.qx;
    else if constexpr (I == 15) s.qy = 1.0f - s.qy;
    else if constexpr (I == 16) s.qx = __builtin_fmaf(__builtin_fabsf(s.zx), s.qx, s.zx);     // g
    else if constexpr (I == 17) s.qy = __builtin_fmaf(__builtin_fabsf(s.zy), s.qy, s.zy);
    else if constexpr (I == 18) { f32x2 g; g.x = s.qx; g.y = s.qy; hi = __builtin_bit_cast(unsigned, __builtin_convertvector(g, h2)); }
    else if constexpr (I == 19) acx_lo_half_lower(lo, s.qx, hi);
    else acx_lo_half_upper(lo, s.qy, hi);
}
// the whole pair at once (epilogues: nothing to interleave with)
__device__ __forceinline__ void gelu3_pair(const GeluK3 k, const float ax, const float ay, unsigned& hi, unsigned& lo) {
    GeluState3 s;
    gelu3_nano<0>(s, k, ax, ay, hi, lo); gelu3_nano<1>(s, k, ax, ay, hi, lo); gelu3_nano<2>(s, k, ax, ay, hi, lo);
    gelu3_nano<3>(s, k, ax, ay, hi, lo); gelu3_nano<4>(s, k, ax, ay, hi, lo); gelu3_nano<5>(s, k, ax, ay, hi, lo);
    gelu3_nano<6>(s, k, ax, ay, hi, lo); gelu3_nano<7>(s, k, ax, ay, hi, lo); gelu3_nano<8>(s, k, ax, ay, hi, lo);
    gelu3_nano<9>(s, k, ax, ay, hi, lo); gelu3_nano<10>(s, k, ax, ay, hi, lo); gelu3_nano<11>(s, k, ax, ay, hi, lo);
    gelu3_nano<12>(s, k, ax, ay, hi, lo); gelu3_nano<13>(s, k, ax, ay, hi, lo); gelu3_nano<14>(s, k, ax, ay, hi, lo);
    gelu3_nano<15>(s, k, ax, ay, hi, lo); gelu3_nano<16>(s, k, ax, ay, hi, lo); gelu3_nano<17>(s, k, ax, ay, hi, lo);
    gelu3_nano<18>(s, k, ax, ay, hi, lo); gelu3_nano<19>(s, k, ax, ay, hi, lo); gelu3_nano<20>(s, k, ax, ay, hi, lo);
}

// The third form in SEVEN micro-steps of 2-4 instructions for a register pair, result left in (s.qx, s.qy) -- the bf16 kernels
// (two pixel tiles or two waves per SIMD: a micro-step per MFMA; no hi / lo split, the caller packs to bf16).  With
// k = gelu_k3(1, 1) the unit is z = 0.5 v and the result gelu(v) itself.
template <int STEP>
__device__ __forceinline__ void gelu3_micro(GeluState3& s, const GeluK3 k, const float ax, const float ay) {
    if constexpr (STEP == 0) { s.zx = ax * k.zs; s.zy = ay * k.zs; }
    else if constexpr (STEP == 1) { s.qx = __builtin_fmaf(__builtin_fabsf(s.zx), k.k4, k.k3); s.qy = __builtin_fmaf(__builtin_fabsf(s.zy), k.k4, k.k3); }
    else if constexpr (STEP == 2) { s.qx = __builtin_fmaf(s.qx, __builtin_fabsf(s.zx), k.k2); s.qy = __builtin_fmaf(s.qy, __builtin_fabsf(s.zy), k.k2); }
    else if constexpr (STEP == 3) { s.qx = __builtin_fmaf(s.qx, __builtin_fabsf(s.zx), k.k1); s.qy = __builtin_fmaf(s.qy, __builtin_fabsf(s.zy), k.k1); }
    else if constexpr (STEP == 4) { s.qx = __builtin_fmaf(s.qx, __builtin_fabsf(s.zx), k.k0); s.qy = __builtin_fmaf(s.qy, __builtin_fabsf(s.zy), k.k0); }
    else if constexpr (STEP == 5) {
        s.qx *= __builtin_fabsf(s.zx); s.qy *= __builtin_fabsf(s.zy);
        s.qx = __builtin_amdgcn_exp2f(s.qx); s.qy = __builtin_amdgcn_exp2f(s.qy);
    } else {
        s.qx = 1.0f - s.qx; s.qy = 1.0f - s.qy;
        s.qx = __builtin_fmaf(__builtin_fabsf(s.zx), s.qx, s.zx); s.qy = __builtin_fmaf(__builtin_fabsf(s.zy), s.qy, s.zy);
    }
}
// ---- the GELU of the FUSED bf16 kernels (round 5): degree 2, and z delivered by the matrix pipe -----------------------------------
// The hidden activation of the bf16 arithmetics is rounded to bf16 (8 significant bits) the moment it exists; a GELU good to 5e-7 is
// three instructions per element more than that needs.  Q of degree 2 (minimax fit of 0.5 |v| (E - erfc) over [0, 9],
// tools/lab/fit_gelu.py 2): |gelu error| <= 8.6e-5 absolute -- below half a bf16 step of the result wherever |gelu| > 0.044,
// an absolute 8.6e-5 below that -- leading coefficient positive, so no clamp, E(0) = 1 exactly.  And the packers store 0.5 W1 (exact
// in bf16) and the kernels stage 0.5 b1: the accumulator IS z = 0.5 v, the scaling step is gone.  Five vector instructions and one
// transcendental per element instead of eight and one; same seven-step interface as gelu3_micro (steps 4-6 are empty), result in
// (s.qx, s.qy).  ax / ay are read by steps 0-3.
constexpr bool kBf16FusedHalfW1 = true;       // api.hip packs wstream_b with 0.5 W1; the fused bf16 kernels stage 0.5 b1
__device__ __forceinline__ GeluK3 gelu_k2h() {
    GeluK3 k;
    k.zs = 1.0f; k.k3 = 0.f; k.k4 = 0.f;
    k.k0 = -1.140932559967041f * 2.0f; k.k1 = -0.4882272183895111f * 4.0f; k.k2 = -0.027643846347928047f * 8.0f;    // -c_j / 0.5^(j+1): the unit is z
    return k;
}
template <int STEP>
__device__ __forceinline__ void gelu2h_micro(GeluState3& s, const GeluK3 k, const float ax, const float ay) {
    if constexpr (STEP == 0) { s.qx = __builtin_fmaf(__builtin_fabsf(ax), k.k2, k.k1); s.qy = __builtin_fmaf(__builtin_fabsf(ay), k.k2, k.k1); }
    else if constexpr (STEP == 1) { s.qx = __builtin_fmaf(s.qx, __builtin_fabsf(ax), k.k0); s.qy = __builtin_fmaf(s.qy, __builtin_fabsf(ay), k.k0); }
    else if constexpr (STEP == 2) {
        s.qx *= __builtin_fabsf(ax); s.qy *= __builtin_fabsf(ay);
        s.qx = __builtin_amdgcn_exp2f(s.qx); s.qy = __builtin_amdgcn_exp2f(s.qy);
    } else if constexpr (STEP == 3) {
        s.qx = 1.0f - s.qx; s.qy = 1.0f - s.qy;
        s.qx = __builtin_fmaf(__builtin_fabsf(ax), s.qx, ax); s.qy = __builtin_fmaf(__builtin_fabsf(ay), s.qy, ay);
    }
}
// one value, all at once (epilogues of the bf16 GEMMs)
__device__ __forceinline__ float gelu3_unit(const float v) {
    const float z = 0.5f * v, a = __builtin_fabsf(z);
    float q = __builtin_fmaf(a, -0.0004882981302216649f * 32.0f, 0.00719997426494956f * 16.0f);
    q = __builtin_fmaf(q, a, -0.05214935168623924f * 8.0f);
    q = __builtin_fmaf(q, a, -0.4595935642719269f * 4.0f);
    q = __builtin_fmaf(q, a, -1.1510010957717896f * 2.0f);
    const float e = __builtin_amdgcn_exp2f(q * a);
    return __builtin_fmaf(a, 1.0f - e, z);
}

}  // namespace acx
