// Pure-register victims, one instruction class each (inline asm so that the class is what runs): which VALU
// operations return wrong results while a split GEMM (back-to-back independent fp16 MFMAs) shares the SIMD?
// Every thread iterates its class `rounds` times on register values and writes the final bits; two launches of the
// same inputs must agree bit for bit.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CLS>
__global__ __launch_bounds__(256) void cls_kernel(const float* __restrict__ seed, unsigned* __restrict__ out, int rounds,
                                                   unsigned long long sm /* (m, -m) */, unsigned long long sk /* (k, -k) */, float smf, float skf) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    float a = seed[(gid * 4) & 0xffff], b = seed[(gid * 4 + 1) & 0xffff], c = seed[(gid * 4 + 2) & 0xffff], d = seed[(gid * 4 + 3) & 0xffff];
    const float m = 0.999f, k = 1e-3f;
    f2 x; x.x = a; x.y = b;
    f2 y; y.x = c; y.y = d;
    f2 mm; mm.x = m; mm.y = -m;
    f2 kk; kk.x = k; kk.y = -k;
    double da = (double)a, db = (double)b;
    unsigned long long ua = __float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    unsigned ub = __float_as_uint(c) | 1u;
    for (int it = 0; it < rounds; ++it) {
        if (CLS == 0) {          // v_fma_f32 (32-bit results)
            asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %1, %3, %0\n\tv_fma_f32 %1, %0, %3, %1"
                         : "+v"(a), "+v"(b) : "v"(m), "v"(k));
        } else if (CLS == 1) {   // v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\tv_pk_fma_f32 %0, %1, %3, %0\n\tv_pk_fma_f32 %1, %0, %3, %1"
                         : "+v"(x), "+v"(y) : "v"(mm), "v"(kk));
        } else if (CLS == 2) {   // v_pk_mul_f32 + v_pk_add_f32
            asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_pk_add_f32 %0, %0, %3\n\tv_pk_mul_f32 %1, %1, %2\n\tv_pk_add_f32 %1, %1, %0"
                         : "+v"(x), "+v"(y) : "v"(mm), "v"(kk));
        } else if (CLS == 3) {   // v_fma_f64 (64-bit result, full-rate on CDNA)
            asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %0, %1, %3, %0\n\tv_fma_f64 %1, %0, %3, %1"
                         : "+v"(da), "+v"(db) : "v"((double)m), "v"((double)k));
        } else if (CLS == 4) {   // 64-bit integer results: v_mad_u64_u32, v_lshlrev_b64
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_lshlrev_b64 %0, 1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0"
                         : "+v"(ua) : "v"(ub) : "vcc");
        } else if (CLS == 5) {   // v_mul_f32 / v_add_f32 / v_mov_b32 mix (32-bit)
            asm volatile("v_mul_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %2\n\tv_add_f32 %1, %1, %0"
                         : "+v"(a), "+v"(b) : "v"(m), "v"(k));
        } else if (CLS == 6) {   // v_pk_mov_b32 / v_mov_b64-style 64-bit moves + v_pk_add_f32
            asm volatile("v_pk_mov_b32 %1, %0, %0 op_sel:[1,0]\n\tv_pk_add_f32 %0, %1, %2\n\tv_pk_mov_b32 %1, %0, %0 op_sel:[1,0]\n\tv_pk_mul_f32 %0, %1, %3"
                         : "+v"(x), "+v"(y) : "v"(kk), "v"(mm));
        } else if (CLS == 8) {   // v_pk_mul_f32 / v_pk_add_f32 with SGPR-pair operands, no modifiers
            asm volatile("v_pk_mul_f32 %0, %0, %2\n\tv_pk_add_f32 %0, %0, %3\n\tv_pk_mul_f32 %1, %1, %2\n\tv_pk_add_f32 %1, %1, %0"
                         : "+v"(x), "+v"(y) : "s"(sm), "s"(sk));
        } else if (CLS == 9) {   // v_pk_fma_f32, VGPR operands, op_sel / neg modifiers as the compiler uses them for complex products
            asm volatile("v_pk_mul_f32 %1, %0, %2 op_sel_hi:[1,0]\n\t"
                         "v_pk_fma_f32 %0, %0, %3, %1 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                         "v_pk_fma_f32 %1, %0, %3, %1 op_sel:[0,0,1] op_sel_hi:[1,0,0]\n\t"
                         "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]"
                         : "+v"(x), "+v"(y) : "v"(mm), "v"(kk));
        } else if (CLS == 10) {  // v_fma_f32 / v_mul_f32 with 32-bit SGPR operands
            asm volatile("v_fma_f32 %0, %0, %2, %1\n\tv_mul_f32 %1, %3, %1\n\tv_fma_f32 %1, %0, %3, %1\n\tv_mul_f32 %0, %2, %0"
                         : "+v"(a), "+v"(b) : "s"(smf), "s"(skf));
        } else if (CLS == 11) {  // the compiled complex product: SGPR-pair operands AND modifiers
            asm volatile("v_pk_mul_f32 %1, %0, %2 op_sel_hi:[1,0]\n\t"
                         "v_pk_fma_f32 %0, %0, %3, %1 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                         "v_pk_fma_f32 %1, %0, %3, %1 op_sel:[0,0,1] op_sel_hi:[1,0,0]\n\t"
                         "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]"
                         : "+v"(x), "+v"(y) : "s"(sm), "s"(sk));
        } else if (CLS == 7) {   // packed fp16: v_pk_fma_f16
            unsigned p = __float_as_uint(a), q = __float_as_uint(b);
            asm volatile("v_pk_fma_f16 %0, %0, %2, %3\n\tv_pk_fma_f16 %1, %1, %2, %0\n\tv_pk_fma_f16 %0, %1, %2, %3\n\tv_pk_fma_f16 %1, %0, %2, %1"
                         : "+v"(p), "+v"(q) : "v"(0x3bff3bffu), "v"(0x14001400u));
            a = __uint_as_float(p); b = __uint_as_float(q);
        }
    }
    unsigned r = __float_as_uint(a) ^ (__float_as_uint(b) * 3u) ^ __float_as_uint(x.x) ^ (__float_as_uint(x.y) * 5u) ^ __float_as_uint(y.x) ^
                 (__float_as_uint(y.y) * 7u) ^ (unsigned)__double_as_longlong(da) ^ (unsigned)(__double_as_longlong(da) >> 32) ^
                 (unsigned)__double_as_longlong(db) ^ (unsigned)(__double_as_longlong(db) >> 29) ^ (unsigned)ua ^ (unsigned)(ua >> 32);
    out[gid] = r;
}

extern "C" int cls_launch(int cls, const float* seed, unsigned* out, int blocks, int rounds, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const float m = 0.999f, k = 1e-3f, nm = -m, nk = -k;
    unsigned um, unm, uk, unk;
    memcpy(&um, &m, 4); memcpy(&unm, &nm, 4); memcpy(&uk, &k, 4); memcpy(&unk, &nk, 4);
    const unsigned long long sm = um | ((unsigned long long)unm << 32), sk = uk | ((unsigned long long)unk << 32);
#define ARGS seed, out, rounds, sm, sk, m, k
    switch (cls) {
        case 0: cls_kernel<0><<<blocks, 256, 0, s>>>(ARGS); break;
        case 1: cls_kernel<1><<<blocks, 256, 0, s>>>(ARGS); break;
        case 2: cls_kernel<2><<<blocks, 256, 0, s>>>(ARGS); break;
        case 3: cls_kernel<3><<<blocks, 256, 0, s>>>(ARGS); break;
        case 4: cls_kernel<4><<<blocks, 256, 0, s>>>(ARGS); break;
        case 5: cls_kernel<5><<<blocks, 256, 0, s>>>(ARGS); break;
        case 6: cls_kernel<6><<<blocks, 256, 0, s>>>(ARGS); break;
        case 7: cls_kernel<7><<<blocks, 256, 0, s>>>(ARGS); break;
        case 8: cls_kernel<8><<<blocks, 256, 0, s>>>(ARGS); break;
        case 9: cls_kernel<9><<<blocks, 256, 0, s>>>(ARGS); break;
        case 10: cls_kernel<10><<<blocks, 256, 0, s>>>(ARGS); break;
        case 11: cls_kernel<11><<<blocks, 256, 0, s>>>(ARGS); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ---- minimal aggressors: register-only MFMA loops ------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// NACC independent accumulators (1 = fully dependent chain, 4 = back-to-back independent as in gemm_split)
template <int NACC>
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
    h8 a[4], b[4];
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = hashu(threadIdx.x * 64 + s * 8 + i + blockIdx.x * 7919);
            a[s][i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f));
            b[s][i] = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.f));
        }
    f16v c[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u + n) & 3], c[n], 0, 0, 0);
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
    if (s == 123.456f) out[threadIdx.x] = s;
}
// live operands: 8 + 8 distinct random fragment sets, 4 independent accumulators: successive MFMAs never see the same
// operand registers (as in a GEMM main loop), still no memory or LDS instruction in the loop
__global__ __launch_bounds__(256) void burn_live(float* out, int iters) {
    h8 a[8], b[8];
    for (int s = 0; s < 8; ++s)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = hashu(threadIdx.x * 128 + s * 8 + i + blockIdx.x * 7919);
            a[s][i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f));
            b[s][i] = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.f));
        }
    f16v c[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 3 * n) & 7], b[(5 * u + n) & 7], c[n], 0, 0, 0);
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
    if (s == 123.456f) out[threadIdx.x] = s;
}
extern "C" int burn_launch(int nacc, float* out, int blocks, int iters, void* stream) {
    if (nacc == 8) { burn_live<<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters); return (int)hipGetLastError(); }
    hipStream_t s = (hipStream_t)stream;
    if (nacc == 1) burn<1><<<blocks, 256, 0, s>>>(out, iters);
    else if (nacc == 4) burn<4><<<blocks, 256, 0, s>>>(out, iters);
    else if (nacc == 6) burn<6><<<blocks, 256, 0, s>>>(out, iters);
    else return -1;
    return (int)hipGetLastError();
}
