"""bf16 precision modes (BASELINE configs[2] "bf16 with fp32 LayerNorm") on the GPU, every test in both:
  "bf16"   bf16 MFMA operands, activations in HBM stay fp32;
  "bf16a"  additionally the activations of stages 0-2 (residual stream, depthwise-conv output) are STORED as bf16 -- SURVEY 8d's
           reading of configs[2] (107.5 MB of activation traffic per clip); arithmetic between load and store unchanged.

The 1e-3 parity bar belongs to the fp32 path; here the checks are
  (i)  kernel exactness: one block / one downsample layer against an emulation of the same arithmetic built from
       the oracle's pieces -- operands rounded to bf16 exactly where the kernels round them, accumulation in
       fp64 -- so what is left is fp32 accumulation order and the odd rounding-boundary flip;
  (ii) model-level drift against the fp32 golden vectors of the reference class, with the tolerance bf16 operands
       (8 mantissa bits) through 18 blocks allow, and the same decisions at the demo's 0.25 threshold.
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from audioset_convnext_inf_amd import _ffi
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
import parity_floor

pytestmark = pytest.mark.gpu

DIMS = (96, 192, 384, 768)
EMU_TOL = 4e-3          # vs the bf16 emulation: activations O(1), one bf16 ulp of a hidden value is 2^-8 relative
DRIFT_LAYER_TOL = 6e-2  # one block vs the fp32 tap
DRIFT_E2E_TOL = 0.25    # logits vs the fp32 goldens (logit range is about +-10)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.fixture(scope="module", params=["bf16", "bf16a"])
def model16(synth_sd, request):
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                      use_speed_perturb=False)
    m.load_state_dict(synth_sd)
    return m.to("cuda").eval().set_precision(request.param)


@pytest.fixture(scope="module")
def ctx16(model16):
    return model16.native_context(torch.device("cuda", 0))


@pytest.fixture(scope="module")
def taps(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_taps.npz"))
    return {k: torch.from_numpy(g[k]) for k in g.files}


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def sp():
    return _ffi.stream_ptr(torch.device("cuda", 0))


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def ln_plain(x, C):
    return F.layer_norm(x, (C,), None, None, 1e-6)


@pytest.mark.parametrize("rows", [None, 3])
@pytest.mark.parametrize("s", [0, 1, 2, 3])
def test_block_bf16(ctx16, model16, taps, synth_sd, s, rows):
    _block_check(ctx16, model16, taps, synth_sd, s, rows)


def _block_check(ctx16, model16, taps, synth_sd, s, rows):
    from oracle import ref_cpu
    C = DIMS[s]
    p = "stages.%d.0." % s
    x0 = taps["ds%d" % s]                                  # NCHW fp32
    if rows is not None:                                   # 3 x W pixels of one clip: not a multiple of any kernel's row tile
        x0 = x0[:1, :, :rows, :].contiguous()
    x_in = x0
    act = model16.precision == "bf16a" and s < 3           # stored tensors of this stage are bf16: x in, y, x out
    if act:
        x0 = x0.to(torch.bfloat16).float()                 # what the block reads
    sd_dw = synth_sd
    if act:                                                # the matrix-pipe depthwise kernel multiplies bf16 weights (dwconv_mfma.hip)
        sd_dw = dict(synth_sd)
        sd_dw[p + "dwconv.weight"] = synth_sd[p + "dwconv.weight"].to(torch.bfloat16).float()
    y = ref_cpu.block_dwconv(sd_dw, s, 0, x0).permute(0, 2, 3, 1)
    if act:
        y = y.to(torch.bfloat16).float()                   # the depthwise conv's output as stored
    # the arithmetic of the bf16 block kernels (mlp_fused_wide_bf16.hip; stage 3: run_mlp_bf16 in api.hip): folds in
    # fp32/fp64, operands rounded to bf16, wide accumulation
    yn = bf(ln_plain(y, C))
    w1 = bf((synth_sd[p + "pwconv1.weight"].double() * synth_sd[p + "norm.weight"].double()[None, :]).float())
    b1 = synth_sd[p + "pwconv1.bias"].double() + synth_sd[p + "pwconv1.weight"].double() @ synth_sd[p + "norm.bias"].double()
    h = bf(F.gelu((yn @ w1.T + b1).float()))
    g = synth_sd[p + "gamma"].double()
    w2 = bf((g[:, None] * synth_sd[p + "pwconv2.weight"].double()).float())
    ref = x0.permute(0, 2, 3, 1).double() + h @ w2.T + g * synth_sd[p + "pwconv2.bias"].double()
    if act:
        ref = ref.float().to(torch.bfloat16).double()      # ... and what it writes

    x = nhwc(x_in)
    B, H, W, _ = x.shape
    need = ctypes.c_size_t()
    _ffi.check(_ffi.lib().acx_block_scratch_bytes(s, B, H, W, ctypes.byref(need)))
    scratch = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    _ffi.check(_ffi.lib().acx_block(ctx16.handle, s, 0, _ffi.ptr(x), B, H, W, _ffi.ptr(scratch), need.value, sp()))
    torch.cuda.synchronize()
    d_emu = maxdiff(x, ref)
    # bf16a: a stored value may round the other way when the fp32 sum sits at a rounding boundary -- one bf16 ulp of the result
    assert d_emu < (EMU_TOL if not act else max(EMU_TOL, 2.0 ** -7 * float(ref.abs().max())))
    if rows is None:
        d_f32 = maxdiff(x.permute(0, 3, 1, 2), taps["s%d.b0.out" % s])
        print("stage %d: vs bf16 emulation %.3g, vs fp32 tap %.3g" % (s, d_emu, d_f32))
        assert d_f32 < DRIFT_LAYER_TOL


def _dw_bf16(ctx, s, j, xb, B, H, W):
    y = torch.empty_like(xb)
    _ffi.check(_ffi.lib().acx_dwconv7_bf16(ctx.handle, s, j, _ffi.ptr(xb), _ffi.ptr(y), B, H, W, sp()))
    torch.cuda.synchronize()
    return y


# (B, H): one short clip; clip boundaries inside a four-row step; a batch that ends inside a segment; segments of an odd number of
# steps; the product shape of the stage at 3 clips
@pytest.mark.parametrize("B,H", [(1, 5), (3, 9), (2, 31), (7, 50), (5, 63), (3, None)])
@pytest.mark.parametrize("s", [0, 1, 2])
def test_dwconv_matrix_kernel(ctx16, model16, synth_sd, s, B, H):
    """acx_dwconv7_bf16 (dwconv_mfma.hip: the depthwise conv of bf16 activations as 4x4x4 bf16 MFMAs, one channel per block)
    against Block.dwconv (convnext.py:58-60,76) on the same bf16 inputs and bf16-rounded weights, accumulated in fp64: the kernel's
    products are exact and its accumulation fp32, so a result may differ from the rounded exact sum by ONE bf16 step where the
    sum sits at a rounding boundary -- and by nothing more, at any shape (clip boundaries, segment ends, image edges)."""
    if model16.precision != "bf16a":
        rc = _ffi.lib().acx_dwconv7_bf16(ctx16.handle, s, 0, None, None, 1, 4, 56 >> s, sp())
        assert rc != 0                                     # no bf16 activations at this precision: refused, not computed
        return
    C, W = DIMS[s], 56 >> s
    H = H if H is not None else 252 >> s
    g = torch.Generator().manual_seed(4000 + 100 * s + B + H)
    x = (torch.randn(B, H, W, C, generator=g) * 2.0).to(torch.bfloat16)
    p = "stages.%d.1." % s
    w = synth_sd[p + "dwconv.weight"].to(torch.bfloat16).double()
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w, synth_sd[p + "dwconv.bias"].double(), padding=3, groups=C).permute(0, 2, 3, 1)
    y = _dw_bf16(ctx16, s, 1, x.cuda().contiguous(), B, H, W).cpu()
    again = _dw_bf16(ctx16, s, 1, x.cuda().contiguous(), B, H, W).cpu()
    assert torch.equal(y.view(torch.int16), again.view(torch.int16))
    # y = bf16(fp32 sum), fp32 sum = exact + e with |e| <= 2^-20 sum |x w| (49 products, each added once, fp32 partial sums) and
    # rounding to bf16 moves a value by at most 2^-8 of itself: |y - exact| <= 2^-8 |exact| + (1 + 2^-8) |e|
    mag = F.conv2d(x.double().abs().permute(0, 3, 1, 2), w.abs(), synth_sd[p + "dwconv.bias"].double().abs(), padding=3, groups=C).permute(0, 2, 3, 1)
    err = (y.double() - ref).abs()
    bound = 2.0 ** -8 * ref.abs() + 1.01 * 2.0 ** -20 * mag
    worst = float((err / bound).max())
    exact = float((y == ref.to(torch.bfloat16)).double().mean())
    print("stage %d B=%d H=%d: %.4f %% of the outputs are the rounded exact sum, worst error %.3f of the bound" % (s, B, H, 100 * exact, worst))
    assert worst <= 1.0 and exact > 0.99


def test_dwconv_matrix_kernel_vs_column_kernel(ctx16, model16, synth_sd):
    """ACX_DW_MFMA = 0 sends bf16 activations through the column / tile kernels (fp32 weights, 49 FMAs): the two arithmetics differ by the
    bf16 rounding of the weights only -- 2^-9 relative per tap, far inside the activations' own rounding."""
    if model16.precision != "bf16a":
        pytest.skip("bf16 activations only")
    refresh = _ffi.lib().acx_tuning_refresh
    s, B = 1, 4
    C, W, H = DIMS[s], 56 >> s, 252 >> s
    x = (torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(77)) * 2.0).to(torch.bfloat16).cuda()
    ym = _dw_bf16(ctx16, s, 0, x, B, H, W).float()
    os.environ["ACX_DW_MFMA"] = "0"
    refresh()
    try:
        yc = _dw_bf16(ctx16, s, 0, x, B, H, W).float()
    finally:
        del os.environ["ACX_DW_MFMA"]
        refresh()
    d = maxdiff(ym, yc)
    print("matrix vs column kernel: max abs diff %.3g at |y| <= %.3g" % (d, float(yc.abs().max())))
    assert d <= 2.0 ** -6 * float(yc.abs().max())


def test_dwconv_matrix_kernel_segmentation_is_invisible(ctx16, model16):
    """The matrix-pipe depthwise kernel cuts the stacked batch into segments by the number of waves it is asked to fill
    (ACX_DWM_WAVES per CU; 8 by default, halved per sub-batch in flight): an output's arithmetic -- bias, then kernel rows 0-3 of its
    tile's X phase, then 4-6 of its Y phase, input quads left to right -- must not depend on where the segments fall."""
    if model16.precision != "bf16a":
        pytest.skip("bf16 activations only")
    refresh = _ffi.lib().acx_tuning_refresh
    s, B = 2, 5
    C, W, H = DIMS[s], 56 >> s, 252 >> s
    x = (torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(99)) * 2.0).to(torch.bfloat16).cuda()
    ref = _dw_bf16(ctx16, s, 2, x, B, H, W)
    try:
        for v in ("2", "5", "9"):
            os.environ["ACX_DWM_WAVES"] = v
            refresh()
            got = _dw_bf16(ctx16, s, 2, x, B, H, W)
            assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), v
    finally:
        os.environ.pop("ACX_DWM_WAVES", None)
        refresh()


@pytest.mark.parametrize("i", [1, 2, 3])
def test_downsample_bf16(ctx16, taps, synth_sd, i):
    Ci, Co = DIMS[i - 1], DIMS[i]
    p = "downsample_layers.%d." % i
    x0 = taps["stage%d" % (i - 1)]
    xn = bf(ln_plain(x0.permute(0, 2, 3, 1), Ci)).permute(0, 3, 1, 2)
    cw = synth_sd[p + "1.weight"].double()
    w = bf((cw * synth_sd[p + "0.weight"].double()[None, :, None, None]).float())
    b = synth_sd[p + "1.bias"].double() + (cw * synth_sd[p + "0.bias"].double()[None, :, None, None]).sum(dim=(1, 2, 3))
    ref = F.conv2d(xn, w, b, stride=2)

    x = nhwc(x0)
    B, H, W, _ = x.shape
    out = torch.empty(B, H // 2, W // 2, Co, device="cuda")
    scratch = torch.empty_like(x)
    _ffi.check(_ffi.lib().acx_downsample(ctx16.handle, i, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scratch), B, H, W, sp()))
    torch.cuda.synchronize()
    d_emu = maxdiff(out.permute(0, 3, 1, 2), ref)
    d_f32 = maxdiff(out.permute(0, 3, 1, 2), taps["ds%d" % i])
    print("downsample %d: vs bf16 emulation %.3g, vs fp32 tap %.3g" % (i, d_emu, d_f32))
    assert d_emu < EMU_TOL
    assert d_f32 < DRIFT_LAYER_TOL


def test_e2e_bf16_drift_on_demo_clip(model16, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_demo.npz"))
    wav = torch.from_numpy(g["pcm16"].astype(np.float32) / 32768.0)[None].cuda()
    out = model16(wav)
    d_logit = maxdiff(out["clipwise_logits"], torch.from_numpy(g["logits"]))
    d_prob = maxdiff(out["clipwise_output"], torch.from_numpy(g["probs"]))
    d_scene = maxdiff(model16.forward_scene_embeddings(wav), torch.from_numpy(g["scene"]))
    print("bf16 e2e drift: logits %.3g probs %.3g scene %.3g" % (d_logit, d_prob, d_scene))
    # contract: the drift bf16 operands allow through 18 blocks; regression bar: 3 x the drift recorded for these rounding points
    parity_floor.check("bf16/g1_demo_drift/" + model16.precision, {"logits": d_logit, "probs": d_prob, "scene": d_scene},
                       {"logits": DRIFT_E2E_TOL, "probs": 0.05, "scene": DRIFT_E2E_TOL}, factor=3.0)
    ref_dec = g["probs"][0] > 0.25
    got_dec = out["clipwise_output"][0].cpu().numpy() > 0.25
    if model16.precision == "bf16":
        assert np.array_equal(ref_dec, got_dec)
    else:           # bf16a: a class whose probability sits within the drift of the threshold may flip -- at most 0.5 % of them
        assert float((ref_dec == got_dec).mean()) >= 0.995, int((ref_dec != got_dec).sum())
    # ranking of the confident classes survives
    top_ref = np.argsort(-g["logits"][0])[:5]
    top_got = np.argsort(-out["clipwise_logits"][0].cpu().numpy())[:5]
    assert set(top_ref[:3]) <= set(top_got)


def test_bf16_label_agreement_on_goldens(model16, golden_dir):
    """Decisions at the demo's 0.25 threshold against the fp32 goldens of the reference class (g2_taps: seeded clips, g2_edge:
    noise / silence / square / sweep): >= 99.5 % of the (clip, class) decisions agree."""
    agree = total = 0
    for name in ("g2_taps.npz", "g2_edge.npz"):
        g = np.load(os.path.join(golden_dir, name))
        wkeys = [k for k in g.files if k == "wav" or k.startswith("wav_")]
        for wk in wkeys:
            pk = "probs" if wk == "wav" else "probs_" + wk[4:]
            if pk not in g.files:
                continue
            wav = torch.from_numpy(g[wk]).cuda()
            if wav.dim() == 1:
                wav = wav[None]
            got = model16(wav)["clipwise_output"].cpu().numpy() > 0.25
            ref = g[pk] > 0.25
            agree += int((got == ref).sum())
            total += ref.size
    assert total > 0
    print("label agreement at 0.25: %d / %d = %.4f" % (agree, total, agree / total))
    assert agree / total >= 0.995


def test_bf16_batch_rows_independent(model16):
    from audioset_convnext_inf_amd import synth
    wav = synth.synth_waveforms(18, 64000, seed=5).cuda()
    out = model16(wav)["clipwise_logits"]
    assert bool(torch.isfinite(out).all())
    solo = model16(wav[11:12])["clipwise_logits"]
    assert torch.equal(solo[0], out[11])


def test_bf16_ragged_clip_lengths(synth_sd):
    """Clip lengths whose pixel counts are not multiples of the kernels' row tiles (32 / 128 / 256 rows), through the whole
    forward (fused block kernels incl. the last block of a stage writing the downsample conv's operand rows): finite,
    batch == solo bit for bit, and within the bf16 drift of the fp32-grade result."""
    from audioset_convnext_inf_amd import synth
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                      use_speed_perturb=False)
    m.load_state_dict(synth_sd)
    m = m.to("cuda").eval()
    for B, L in ((3, 43840), (1, 30000), (5, 52000)):
        wav = synth.synth_waveforms(B, L, seed=B).cuda()
        m.set_precision("fp32_split")
        ref = m(wav)["clipwise_logits"].clone()
        ref_fr = m.forward_frame_embeddings(wav).clone()
        for prec in ("bf16", "bf16a"):
            m.set_precision(prec)
            a = m(wav)["clipwise_logits"].clone()
            fr = m.forward_frame_embeddings(wav).clone()
            assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(fr).all())
            solo = m(wav[B - 1:B])["clipwise_logits"]
            assert torch.equal(solo[0], a[B - 1])
            d, dfr = maxdiff(a, ref), maxdiff(fr, ref_fr)
            print("B=%d L=%d: %s vs fp32_split: logits %.3g frame %.3g" % (B, L, prec, d, dfr))
            assert d < DRIFT_E2E_TOL and dfr < 2 * DRIFT_E2E_TOL


def test_bf16_forward_is_graph_capturable(model16):
    """The bf16 launch path (persistent stage-0 kernel included) neither allocates nor synchronises: capture into a
    hipGraph, replay == eager bit for bit."""
    from audioset_convnext_inf_amd import synth
    wav = synth.synth_waveforms(3, 48000, seed=19).cuda()
    eager = model16(wav)["clipwise_logits"].clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model16(wav)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = model16(wav)["clipwise_logits"]
    torch.cuda.synchronize()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_precision_switch_rebuilds_context(synth_sd):
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                      use_speed_perturb=False)
    m.load_state_dict(synth_sd)
    m = m.to("cuda").eval().set_precision("fp32")
    from audioset_convnext_inf_amd import synth
    wav = synth.synth_waveforms(2, 32000, seed=9).cuda()
    a = m(wav)["clipwise_logits"].clone()
    b = m.set_precision("bf16")(wav)["clipwise_logits"].clone()
    c = m.set_precision("fp32")(wav)["clipwise_logits"].clone()
    assert torch.equal(a, c)
    assert not torch.equal(a, b) and maxdiff(a, b) < DRIFT_E2E_TOL
    d = m.set_precision("bf16a")(wav)["clipwise_logits"].clone()
    assert not torch.equal(b, d) and maxdiff(a, d) < DRIFT_E2E_TOL
    with pytest.raises(ValueError):
        m.set_precision("fp8")
