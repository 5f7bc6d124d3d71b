"""The number format behind the default arithmetic (csrc/gemm_split.hip), checked on the CPU:
an fp32 value carried as fp16 hi + fp16 lo is represented to 2^-23 relative in the worst case (exact for ~40 % of random
mantissas; fp32 itself rounds to 2^-24), and hi*hi + hi*lo + lo*hi with fp32
accumulation is as close to the exact product sum as an fp32 FMA chain is.  (The kernels themselves are tested on the
GPU; this pins the error analysis the design rests on.)"""
import numpy as np
import pytest
import torch


def split16(x, scale):
    """hi, lo as float32 tensors holding exactly representable fp16 values of x * scale."""
    v = x * scale
    hi = v.to(torch.float16).to(torch.float32)
    lo = (v - hi).to(torch.float16).to(torch.float32)
    return hi, lo


def pow2_scale(x):
    mx = float(x.abs().max())
    return 2.0 ** (14 - int(np.floor(np.log2(mx))))


@pytest.mark.parametrize("scale_in", [1.0, 1e-3, 27.0])
def test_two_fp16_halves_keep_fp32_grade_precision(scale_in):
    torch.manual_seed(0)
    x = torch.randn(1 << 16) * scale_in
    s = pow2_scale(x)
    hi, lo = split16(x, s)
    rec = (hi.double() + lo.double()) / s
    rel = ((rec - x.double()).abs() / x.double().abs().clamp_min(1e-30))
    big = x.abs() > float(x.abs().max()) * 2.0 ** -13          # operands within 2^13 of the largest one
    assert float(rel[big].max()) <= 2.0 ** -22                 # <= one fp32 ulp: the representation is not the limit
    assert bool((hi.abs() <= 65504).all())


def test_split_representation_error_bound():
    """The measured bound behind DESIGN.md 3a ('say what the arithmetic is'): over 2^20 values spanning the
    whole range a layer's pre-scale produces -- from the top binade down to where the lo half is an fp16 SUBNORMAL --
    v = hi + lo + e with |e| <= 2^-23 |v| while the remainder v - hi is a normal fp16 number; four in ten values are exact;
    once lo is subnormal the ABSOLUTE error is bounded by half its spacing, 2^-25, of the scaled value."""
    g = torch.Generator().manual_seed(7)
    n = 1 << 20
    # scaled values as the kernels see them: max |v| in [2^14, 2^15) (api.hip s16_scale), magnitudes over 30 binades
    mag = torch.exp2(torch.rand(n, generator=g, dtype=torch.float64) * 30.0 - 15.0)          # 2^-15 .. 2^15
    v = (mag * (1.0 + torch.rand(n, generator=g, dtype=torch.float64))).float() / 2.0
    v = v * torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0)
    hi = v.to(torch.float16).to(torch.float32)
    lo = (v - hi).to(torch.float16).to(torch.float32)
    err = (hi.double() + lo.double() - v.double()).abs()
    rel = err / v.double().abs()
    lo_normal = (v - hi).abs() >= 2.0 ** -14                 # the remainder lies in fp16's normal range
    worst = float(rel[lo_normal].max())
    print("worst relative representation error %.3g = 2^%.2f; exact: %.1f %%" % (worst, np.log2(worst), 100 * float((err == 0).float().mean())))
    assert worst <= 2.0 ** -23 * (1 + 1e-6)
    assert worst > 2.0 ** -24                                # the bound is NOT fp32's 2^-24: do not claim 24 bits
    assert float((err == 0).float().mean()) > 0.3             # (42 % of uniformly random mantissas are carried exactly)
    # lo subnormal (or flushed to zero by the rounding): absolute error <= half the fp16 subnormal spacing 2^-24 of the SCALED
    # value -- 2^-39 of the layer's largest operand
    assert float(err[~lo_normal].max()) <= 2.0 ** -25 * (1 + 1e-9)
    # and the dropped product term: |al bl| <= 2^-22 |a b| (each lo is at most 2^-11 of its value)
    a, b = v[: n // 2], v[n // 2:]
    al, bl = lo[: n // 2], lo[n // 2:]
    ratio = (al.double() * bl.double()).abs() / (a.double() * b.double()).abs()
    assert float(ratio.max()) <= 2.0 ** -22


@pytest.mark.parametrize("K,act_scale", [(96, 2048.0), (384, 2048.0), (1536, 16.0)])
def test_three_term_product_matches_fp32_matmul_error(K, act_scale):
    torch.manual_seed(1)
    M, N = 256, 192
    a = torch.randn(M, K) * (1.0 if act_scale == 2048.0 else 3.0)          # LayerNorm outputs / hidden activations
    w = torch.randn(N, K) * 0.05
    exact = a.double() @ w.double().T
    ws = pow2_scale(w)
    ah, al = split16(a, act_scale)
    wh, wl = split16(w, ws)
    # every partial product of two fp16 numbers is exact in fp32; the MFMA accumulates in fp32
    acc = torch.zeros(M, N)
    for x, y in ((al, wh), (ah, wl), (ah, wh)):
        acc = acc + x @ y.T
    got = acc.double() / (act_scale * ws)
    plain = (a @ w.T).double()
    err_split = float((got - exact).abs().max())
    err_fp32 = float((plain - exact).abs().max())
    scale = float(exact.abs().max())
    print("K=%d: split %.3g, plain fp32 matmul %.3g (relative to max |c| = %.3g)" % (K, err_split / scale, err_fp32 / scale, scale))
    assert err_split <= 2.0 * err_fp32 + 1e-7 * scale
    # dropping the lo halves altogether (plain fp16 operands) is 2-3 orders of magnitude worse: the terms matter
    err_hi_only = float(((ah @ wh.T).double() / (act_scale * ws) - exact).abs().max())
    assert err_hi_only > 50 * err_split


# ---- the GELU of the fused kernels (csrc/split_math.h, third form; DESIGN.md 3a') -------------------------------------------
def _gelu3_constants():
    """The five coefficients as the kernels carry them (gelu_k3 in split_math.h): parsed from the source, so the test pins
    what ships."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audioset-convnext-inf_amd", "csrc",
                            "split_math.h")).read()
    body = src[src.index("GeluK3 gelu_k3("):]
    body = body[:body.index("return k;")]
    c = [float(m) for m in re.findall(r"k\.k[0-4] = (-?[0-9.]+)f \* s;", body)]
    assert len(c) == 5
    return c


def _f32(x):
    with np.errstate(over="ignore"):          # a far-out polynomial value overflows to inf, as it does on the GPU
        return np.asarray(x, np.float64).astype(np.float32).astype(np.float64)


def _gelu3_fp32(v, kh=1.0):
    """gelu3_nano step by step, every instruction's result rounded to fp32 (an FMA = one rounding), in the unit z = 0.5 kh v."""
    h = 0.5 * kh
    k = [c / h ** (j + 1) for j, c in enumerate(_gelu3_constants())]       # exact: h is a power of two
    assert all(abs(x) > 2.0 ** -126 and abs(x) < 2.0 ** 127 for x in k)
    z = _f32(np.asarray(v, np.float64) * h)
    a = np.abs(z)
    q = _f32(a * k[4] + k[3])
    q = _f32(q * a + k[2]); q = _f32(q * a + k[1]); q = _f32(q * a + k[0])
    with np.errstate(over="ignore", under="ignore", invalid="raise"):
        q = _f32(q * a)
        e = _f32(np.exp2(q))
    r = _f32(1.0 - e)
    return _f32(a * r + z) / kh


def test_gelu_third_form_error_bound():
    from scipy.special import erf
    v = np.linspace(-12.0, 12.0, 480001)
    ref = 0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))
    for kh in (1.0, 2.0 ** 12, 2.0 ** -24):                   # the hidden scales acx_finalize can choose (2^-24 .. 2^12)
        g = _gelu3_fp32(v, kh)
        err = np.abs(g - ref)
        assert err.max() <= 1.1e-6, (kh, err.max(), v[err.argmax()])
        small = np.abs(v) < 0.25                              # E(0) = 1 exactly: relative accuracy near zero
        assert (err[small] <= 6e-6 * np.abs(ref[small]) + 1e-12).all()
    # far out: E underflows to 0 (the leading coefficient is positive: no clamp needed), gelu -> v or -0
    far = np.array([20.0, 1e3, 1e10, 1e30])
    assert np.array_equal(_gelu3_fp32(far), _f32(far))
    assert (_gelu3_fp32(-far) == 0.0).all()
