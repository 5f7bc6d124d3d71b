// Shared by the split-fp16 kernels (gemm_split.hip, mlp_fused_split.hip): vector types, the packed-fp32 GELU and
// the fp32 -> (fp16 hi, fp16 lo) split.
#pragma once
#include <hip/hip_runtime.h>

namespace acx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 bc2(float v) { f32x2 r; r.x = v; r.y = v; return r; }

// v = a * s;  g = gelu_erf(v) * kH (A&S 7.1.26, see gemm.hip) for two values, in three pieces; the result comes
// back as packed fp16 hi / lo halves
struct GeluConsts { float ps, cq, ca, cb; };
__device__ __forceinline__ void gelu_piece1(f32x2 a, const GeluConsts k, f32x2& av, f32x2& t, f32x2& e) {
    av.x = __builtin_fabsf(a.x); av.y = __builtin_fabsf(a.y);
    const f32x2 den = fma2(av, bc2(k.ps), bc2(1.0f));
    t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
    const f32x2 u = a * bc2(k.cq);                  // scaled first: a * a alone may overflow for very small weights
    const f32x2 ex = -(u * u);
    e.x = __builtin_amdgcn_exp2f(ex.x); e.y = __builtin_amdgcn_exp2f(ex.y);
}
__device__ __forceinline__ void gelu_piece2(f32x2 a, f32x2 av, f32x2 t, f32x2 e, const GeluConsts k, f32x2& g) {
    f32x2 pl = fma2(t, bc2(1.061405429f), bc2(-1.453152027f));
    pl = fma2(pl, t, bc2(1.421413741f));
    pl = fma2(pl, t, bc2(-0.284496736f));
    pl = fma2(pl, t, bc2(0.254829592f));
    const f32x2 q = pl * t * e;
    f32x2 pos = a * bc2(k.cb);
    pos.x = __builtin_fmaxf(pos.x, 0.f); pos.y = __builtin_fmaxf(pos.y, 0.f);
    g = fma2(av * bc2(k.ca), q, pos);
}
__device__ __forceinline__ void gelu_piece3(f32x2 g, unsigned& hi, unsigned& lo) {
#ifdef ACX_FSLAB_NO_GELU    // diagnostic (tools/mlp_split_lab.hip): (almost) no VALU work between the two products
    hi = __builtin_bit_cast(unsigned, g.x); lo = __builtin_bit_cast(unsigned, g.y);
    return;
#endif
    g.x = __builtin_fminf(g.x, 65504.f); g.y = __builtin_fminf(g.y, 65504.f);
    const h2 h = __builtin_convertvector(g, h2);
    const f32x2 back = __builtin_convertvector(h, f32x2);
    const h2 l = __builtin_convertvector(g - back, h2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

}  // namespace acx
