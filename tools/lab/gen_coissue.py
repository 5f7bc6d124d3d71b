#!/usr/bin/env python3
"""Generates tools/lab/coissue_bench.hip: what does a wave's instruction stream cost when vector / LDS instructions sit in
the gaps between v_mfma_f32_32x32x16_f16 instructions?  (the round-2 version of this bench was built
from C++ statements + sched_barrier; hipcc clumped and SLP-packed them, so it measured something else.)

Every variant is ONE `asm volatile` block per loop iteration -- the instruction order in the binary IS the order written
here -- and `check_coissue.py` disassembles the object and fails unless every variant shows exactly the intended pattern
(N fillers between consecutive MFMAs, no v_pk_*_f32 unless the variant asks for them).

    python3 tools/lab/gen_coissue.py && python3 tools/lab/check_coissue.py        # here (no GPU needed)
    build/labs/coissue_bench > gpurun_out/coissue.txt                               # on the GPU box

Per variant the bench prints shader cycles per MFMA gap (s_memtime around the loop, median over workgroups) and wall ns.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NM = 16          # MFMAs per iteration, two accumulator chains alternating (as the product kernels chain 3 on one)

# filler generators: each returns the list of asm lines for gap `g` (0..NM-1) of the iteration
def fma(n):
    return lambda g: ["v_fma_f32 %%[f%d], %%[f%d], %%[c1], %%[c2]" % ((g * n + i) % 8, (g * n + i) % 8) for i in range(n)]

def mul_exp(n_fma, n_exp):       # n_fma plain + n_exp transcendental (8-cycle) per gap: the GELU's mix
    def f(g):
        out = []
        for i in range(n_fma):
            r = (g * (n_fma + n_exp) + i) % 8
            out.append("v_fma_f32 %%[f%d], %%[f%d], %%[c1], %%[c2]" % (r, r))
        for i in range(n_exp):
            r = (g * (n_fma + n_exp) + n_fma + i) % 8
            out.append("v_exp_f32 %%[f%d], %%[f%d]" % (r, r))
        return out
    return f

def pkfma(n):                    # packed fp32: the anti-lever of MI355X_MICROARCH.md (cycle constants)
    return lambda g: ["v_pk_fma_f32 %%[p%d], %%[p%d], %%[pc], %%[pc]" % ((g * n + i) % 4, (g * n + i) % 4) for i in range(n)]

def dsread(n_rd, n_fma):         # LDS fragment reads (never consumed: the queue's own back-pressure paces them)
    def f(g):
        out = ["ds_read_b128 %%[d%d], %%[la] offset:%d" % ((g * n_rd + i) % 4, ((g * n_rd + i) % 16) * 1024) for i in range(n_rd)]
        for i in range(n_fma):
            r = (g * n_fma + i) % 8
            out.append("v_fma_f32 %%[f%d], %%[f%d], %%[c1], %%[c2]" % (r, r))
        return out
    return f

def cvt_mix(n):                  # the split's instructions: v_cvt_pk_f16_f32 + v_fma_mix_f32
    def f(g):
        out = []
        for i in range(n):
            r = (g * n + i) % 8
            if i % 3 == 0:
                out.append("v_cvt_pk_f16_f32 %%[f%d], %%[f%d], %%[f%d]" % (r, r, (r + 1) % 8))
            else:
                out.append("v_fma_mix_f32 %%[f%d], %%[f%d], 1.0, -%%[f%d] op_sel_hi:[0,0,1]" % (r, r, (r + 1) % 8))
        return out
    return f

def op(fmt, n):                  # n copies per gap of one instruction form; {r} = rotating register f0..f7, {q} = the next one
    def f(g):
        out = []
        for i in range(n):
            r = (g * n + i) % 8
            out.append(fmt.replace("{r}", "%[f" + str(r) + "]").replace("{q}", "%[f" + str((r + 1) % 8) + "]"))
        return out
    return f

OPS = {   # VALU-only issue cost per opcode form, one and two waves per SIMD
    "fma_vvv": "v_fma_f32 {r}, {r}, %%[c1], %%[c2]",
    "fma_sgpr": "v_fma_f32 {r}, {r}, %%[s1], %%[c2]",
    "fma_abs": "v_fma_f32 {r}, |{r}|, %%[c1], %%[c2]",
    "fmaak": "v_fmaak_f32 {r}, {r}, %%[c1], 0x3f7fbe77",
    "mul_e32": "v_mul_f32 {r}, %%[c1], {r}",
    "exp": "v_exp_f32 {r}, {r}",
    "rcp": "v_rcp_f32 {r}, {r}",
    "cvt_pk": "v_cvt_pk_f16_f32 {r}, {r}, {q}",
    "fma_mix": "v_fma_mix_f32 {r}, {r}, 1.0, -{q} op_sel_hi:[0,0,1]",
    "mov": "v_mov_b32 {r}, {q}",
}

# name -> (mfma?, filler fn or None, waves per workgroup, role)   role: 'all' every wave runs the stream;
# 'split' waves 0-3 MFMA only / waves 4-7 fillers only (SIMD partners: waves w and w+4 share a SIMD)
VARIANTS = []
for n in range(0, 9):
    VARIANTS.append(("mfma_fma%d_w4" % n, True, fma(n) if n else None, 4, "all", n, "fma"))
VARIANTS.append(("mfma_fma10_w4", True, fma(10), 4, "all", 10, "fma"))
for n in (0, 2, 4, 6, 8, 10, 12):
    VARIANTS.append(("mfma_fma%d_w8" % n, True, fma(n) if n else None, 8, "all", n, "fma"))
for n in (2, 4, 6, 8):
    VARIANTS.append(("valu_only_fma%d_w4" % n, False, fma(n), 4, "all", n, "fma"))
VARIANTS.append(("valu_only_fma6_w8", False, fma(6), 8, "all", 6, "fma"))
for (a, b) in ((3, 1), (4, 1), (5, 1), (2, 2), (4, 2)):
    VARIANTS.append(("mfma_fma%d_exp%d_w4" % (a, b), True, mul_exp(a, b), 4, "all", a + b, "fma+exp"))
VARIANTS.append(("mfma_fma4_exp1_w8", True, mul_exp(4, 1), 8, "all", 5, "fma+exp"))
for n in (1, 2, 3):
    VARIANTS.append(("mfma_pkfma%d_w4" % n, True, pkfma(n), 4, "all", n, "pk"))
VARIANTS.append(("mfma_pkfma2_w8", True, pkfma(2), 8, "all", 2, "pk"))
for (r, f) in ((1, 0), (2, 0), (1, 4), (2, 3)):
    VARIANTS.append(("mfma_ds%d_fma%d_w4" % (r, f), True, dsread(r, f), 4, "all", r + f, "ds"))
for n in (3, 5):
    VARIANTS.append(("mfma_cvtmix%d_w4" % n, True, cvt_mix(n), 4, "all", n, "mix"))
def unit_pattern(g):              # the wide kernel's unit: gaps 0,1: 3 plain; gap 2: counted wait + two fragment reads + 1 plain
    r = (3 * g) % 8
    if g % 3 == 2:
        return ["s_waitcnt lgkmcnt(2)", "ds_read_b128 %%[d%d], %%[la] offset:%d" % ((2 * (g // 3)) % 4, ((2 * (g // 3)) % 16) * 1024),
                "ds_read_b128 %%[d%d], %%[la] offset:%d" % ((2 * (g // 3) + 1) % 4, ((2 * (g // 3) + 1) % 16) * 1024),
                "v_fma_f32 %%[f%d], %%[f%d], %%[c1], %%[c2]" % (r, r)]
    return ["v_fma_f32 %%[f%d], %%[f%d], %%[c1], %%[c2]" % ((r + i) % 8, (r + i) % 8) for i in range(3)]

def unit_pattern_trans(g):        # the same with the GELU's transcendentals: one v_exp / v_rcp in every plain gap
    out = unit_pattern(g)
    if g % 3 != 2:
        out[1] = "v_exp_f32 %%[f%d], %%[f%d]" % ((3 * g + 1) % 8, (3 * g + 1) % 8)
    return out

def unit_pattern_m0(g):           # + an LDS-DMA-like scalar clump (s_mov m0 / s_nop) in the first gap of every other unit
    out = unit_pattern(g)
    if g % 6 == 0:
        out = ["s_mov_b32 m0, %%[sa]", "s_nop 0", "v_mov_b32 %%[f7], %%[f6]"] + out[:1]
    return out

VARIANTS.append(("chain1_fma4_w4", True, fma(4), 4, "chain1", 4, "fma"))
VARIANTS.append(("chain6_fma4_w4", True, fma(4), 4, "chain6", 4, "fma"))
VARIANTS.append(("unit_w4", True, unit_pattern, 4, "all", 4, "unit"))
VARIANTS.append(("unit_chain1_w4", True, unit_pattern, 4, "chain1", 4, "unit"))
VARIANTS.append(("unit_trans_chain1_w4", True, unit_pattern_trans, 4, "chain1", 4, "unit"))
VARIANTS.append(("unit_m0_chain1_w4", True, unit_pattern_m0, 4, "chain1", 4, "unit"))
for name_, fmt_ in OPS.items():
    for w_ in (4, 8):
        VARIANTS.append(("valu_only_%s_w%d" % (name_, w_), False, op(fmt_, 6), w_, "all", 6, "op"))
# two-segment streams (8 waves): seg 1 = 16 x (MFMA + a fillers), barrier, seg 2 = 16 x b fillers, barrier; the GELU's mix
# (5 plain : 1 transcendental).  'stagger': waves 4-7 run seg 2 first, so a SIMD's two waves are always in different segments;
# 'lockstep': same order in all waves.  uniform reference: 16 x (MFMA + a + b fillers), one segment.
for (a_, b_) in ((4, 4), (3, 5), (5, 3), (2, 6)):
    VARIANTS.append(("seg_stagger_%d_%d_w8" % (a_, b_), True, (mul_exp(a_ - 1, 1) if a_ > 1 else fma(a_), mul_exp(b_ - 1, 1)), 8, "stagger", a_ + b_, "seg"))
    VARIANTS.append(("seg_lockstep_%d_%d_w8" % (a_, b_), True, (mul_exp(a_ - 1, 1) if a_ > 1 else fma(a_), mul_exp(b_ - 1, 1)), 8, "lockstep", a_ + b_, "seg"))
VARIANTS.append(("uniform_fma6_exp2_w8", True, mul_exp(6, 2), 8, "all", 8, "fma+exp"))
VARIANTS.append(("split_mfma_vs_fma6_w8", True, fma(6), 8, "split", 6, "fma"))
VARIANTS.append(("split_mfma_vs_fma12_w8", True, fma(12), 8, "split", 12, "fma"))

MF = "v_mfma_f32_32x32x16_f16 %%[acc%d], %%[a], %%[b], %%[acc%d]"

OPERANDS = (': [acc0] "+v"(acc0), [acc1] "+v"(acc1), [f0] "+v"(f[0]), [f1] "+v"(f[1]), [f2] "+v"(f[2]), [f3] "+v"(f[3]), '
            '[f4] "+v"(f[4]), [f5] "+v"(f[5]), [f6] "+v"(f[6]), [f7] "+v"(f[7]), [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), '
            '[p3] "+v"(p[3]), [d0] "=&v"(d[0]), [d1] "=&v"(d[1]), [d2] "=&v"(d[2]), [d3] "=&v"(d[3])\n'
            '            : [a] "v"(a), [b] "v"(b), [c1] "v"(c1), [c2] "v"(c2), [pc] "v"(pc), [la] "v"(la), [s1] "s"(s1), [sa] "s"(sa) : "memory"')


def asm_block(lines):
    body = "\n".join('            "%s\\n"' % l.replace("%%", "%") for l in lines)
    return "        asm volatile(\n%s\n            %s);\n" % (body, OPERANDS)


def kernel(name, mfma, filler, waves, role):
    if role in ("stagger", "lockstep"):
        f1, f2 = filler
        seg1, seg2 = [], []
        for g in range(NM):
            seg1.append(MF % (g & 1, g & 1)); seg1 += f1(g)
            seg2 += f2(g)
        out = []
        out.append("extern \"C\" __global__ __launch_bounds__(%d) void %s(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {\n" % (waves * 64, name))
        out.append("    PROLOGUE\n")
        out.append("    const bool late = %s;\n" % ("__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4" if role == "stagger" else "false"))
        out.append("    const unsigned long long t0 = stamp();\n")
        out.append("    if (late) {\n" + asm_block(seg2) + "      __builtin_amdgcn_s_barrier();\n    }\n")
        out.append("    for (int it = 0; it < iters; ++it) {\n" + asm_block(seg1) + "      __builtin_amdgcn_s_barrier();\n" + asm_block(seg2) + "      __builtin_amdgcn_s_barrier();\n    }\n")
        out.append("    if (late) {\n" + asm_block(seg1) + "    }\n")
        out.append("    const unsigned long long t1 = stamp();\n")
        out.append("    EPILOGUE\n}\n\n")
        return "".join(out)
    both, only_m, only_v = [], [], []
    for g in range(NM):
        fl = filler(g) if filler else []
        a_ = 0 if role == "chain1" else ((g // 6) & 1 if role == "chain6" else g & 1)
        if mfma:
            both.append(MF % (a_, a_)); only_m.append(MF % (a_, a_))
        both += fl; only_v += fl
    out = []
    out.append("extern \"C\" __global__ __launch_bounds__(%d) void %s(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {\n" % (waves * 64, name))
    out.append("    PROLOGUE\n")
    if role == "split":
        out.append("    const bool matrix_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4;\n")
        out.append("    const unsigned long long t0 = stamp();\n")
        out.append("    if (matrix_wave) {\n      for (int it = 0; it < iters; ++it) {\n" + asm_block(only_m) + "      }\n    } else {\n      for (int it = 0; it < iters; ++it) {\n" + asm_block(only_v) + "      }\n    }\n")
    else:
        out.append("    const unsigned long long t0 = stamp();\n")
        out.append("    for (int it = 0; it < iters; ++it) {\n" + asm_block(both) + "    }\n")
    out.append("    const unsigned long long t1 = stamp();\n")
    out.append("    EPILOGUE\n}\n\n")
    return "".join(out)


HEADER = r'''// GENERATED by tools/lab/gen_coissue.py -- do not edit.  See that file for what this measures.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// live, non-trivial operands from memory (zeros would raise the clock: cdna_hip_programming.md rule 25)
#define PROLOGUE                                                                                              \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                               \
    const int tid = threadIdx.x;                                                                              \
    for (int i = tid; i < 40960; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = in[(i * 7 + blockIdx.x) & 4095]; \
    __syncthreads();                                                                                          \
    f32x16 acc0, acc1;                                                                                        \
    for (int i = 0; i < 16; ++i) { acc0[i] = in[(tid + i) & 4095]; acc1[i] = in[(tid + 16 + i) & 4095]; }     \
    h8 a, b;                                                                                                  \
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(tid * 8 + i) & 4095]; b[i] = (_Float16)in[(tid * 8 + i + 2048) & 4095]; } \
    float f[8];                                                                                               \
    for (int i = 0; i < 8; ++i) f[i] = in[(tid + 64 * i) & 4095];                                             \
    f32x2 p[4];                                                                                               \
    for (int i = 0; i < 4; ++i) { p[i].x = in[(tid + i) & 4095]; p[i].y = in[(tid + 9 * i) & 4095]; }         \
    f32x2 pc; pc.x = 0.999f; pc.y = 0.998f;                                                                   \
    f32x4 d[4];                                                                                               \
    const float c1 = 0.9999f, c2 = 1e-4f;                                                                     \
    const unsigned la = (unsigned)(tid & 63) * 16;                                                            \
    const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c1)));  \
    const unsigned sa = __builtin_amdgcn_readfirstlane(la) & 0xfc00;

#define EPILOGUE                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    float s = 0.f;                                                                                            \
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];                                                      \
    for (int i = 0; i < 8; ++i) s += f[i];                                                                    \
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y + d[i][0] + d[i][3];                                     \
    out[blockIdx.x * blockDim.x + tid] = s;                                                                   \
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;

'''

MAIN = r'''
struct Variant { const char* name; void (*fn)(const float*, float*, unsigned long long*, int); int waves; int fillers; const char* kind; bool mfma; bool split; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int blocks = 256;
    float *in, *out; unsigned long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    {
        std::vector<float> h(4096);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9 & 0xffff) - 32768) / 32768.0f; }
        hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    }
    Variant vs[] = {
VARIANT_TABLE
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# %d MFMAs (v_mfma_f32_32x32x16_f16) per iteration, %d iterations, %d workgroups (one per CU, 160 KB LDS claimed)\n", NM, iters, blocks);
    printf("# cycles = s_memtime ticks of one wave's loop / (iterations x %d): median over waves; ns = wall per gap\n", NM);
    printf("%-28s %6s %8s %10s %12s %10s\n", "variant", "waves", "fillers", "kind", "cyc/gap", "ns/gap");
    for (auto& v : vs) {
        hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int rep = 0; rep < 2; ++rep) {     // first launch warms up
            hipEventRecord(e0, 0);
            v.fn<<<blocks, v.waves * 64, 160 * 1024, 0>>>(in, out, cyc, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
        }
        if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", v.name); continue; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 8);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> c, c2;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < v.waves; ++w) (v.split && w >= 4 ? c2 : c).push_back((double)h[b * 8 + w] / ((double)iters * NM));
        std::sort(c.begin(), c.end());
        printf("%-28s %6d %8d %10s %12.2f %10.2f", v.name, v.waves, v.fillers, v.kind, c[c.size() / 2], ms * 1e6 / ((double)iters * NM));
        if (v.split) { std::sort(c2.begin(), c2.end()); printf("   (filler waves: %.2f cyc/gap)", c2[c2.size() / 2]); }
        printf("\n");
    }
    return 0;
}
'''


def main():
    src = [HEADER, "#define NM %d\n\n" % NM]
    table = []
    for (name, mfma, filler, waves, role, nfill, kind) in VARIANTS:
        src.append(kernel(name, mfma, filler, waves, role))
        table.append('        {"%s", %s, %d, %d, "%s", %s, %s},' % (name, name, waves, nfill, kind, "true" if mfma else "false", "true" if role == "split" else "false"))
    src.append(MAIN.replace("VARIANT_TABLE", "\n".join(table)))
    out = os.path.join(ROOT, "tools", "lab", "coissue_bench.hip")
    with open(out, "w") as f:
        f.write("".join(src))
    # the expected pattern per kernel, for check_coissue.py
    with open(os.path.join(ROOT, "tools", "lab", "coissue_expect.txt"), "w") as f:
        for (name, mfma, filler, waves, role, nfill, kind) in VARIANTS:
            f.write("%s %d %d %s %s\n" % (name, 1 if mfma else 0, nfill, kind, role))
    print("wrote", out)


if __name__ == "__main__":
    main()
