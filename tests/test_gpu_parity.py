"""GPU parity tests proper (`-m gpu`): the HIP path, called through the C ABI (libacx.so via ctypes),
against (i) the committed golden vectors generated from the reference model class and (ii) the CPU
oracle on the same seeded inputs.  Tolerances are written next to each check; the end-to-end bar is
BASELINE.json's 1e-3 abs (fp32)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from audioset_convnext_inf_amd import _ffi, synth
import parity_floor
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny

pytestmark = pytest.mark.gpu

E2E_TOL = 1e-3          # north_star: logits / probs / embeddings within 1e-3 abs (fp32)
LAYER_TOL = 1e-4        # single layer, O(1) activations: fp32 re-association noise only
DIMS = (96, 192, 384, 768)


# Both fp32-grade arithmetic modes face the same checks at the same tolerances: "fp32" multiplies on the
# f32-input matrix cores, "fp32_split" carries every fp32 operand as two fp16 halves (include/acx.h).
@pytest.fixture(scope="module", params=["fp32", "fp32_split"])
def model(synth_sd, request):
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                      use_speed_perturb=False)
    m.load_state_dict(synth_sd)
    return m.to("cuda").eval().set_precision(request.param)


@pytest.fixture(scope="module")
def ctx(model):
    return model.native_context(torch.device("cuda", 0))


@pytest.fixture(scope="module")
def taps(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_taps.npz"))
    return {k: torch.from_numpy(g[k]) for k in g.files}


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def sp():
    return _ffi.stream_ptr(torch.device("cuda", 0))


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def test_library_loaded_is_in_tree():
    assert os.path.samefile(_ffi.LIB_PATH, os.path.join(os.path.dirname(_ffi.__file__), "libacx.so"))
    assert _ffi.lib().acx_version() >= 100


def test_logmel_kernel(ctx, taps):
    wav = taps["wav"].cuda()
    B, L = wav.shape
    T = _ffi.num_frames(L)
    out = torch.empty(B, T, 224, device="cuda")
    _ffi.check(_ffi.lib().acx_logmel_bn0(ctx.handle, _ffi.ptr(wav), B, L, _ffi.ptr(out), 0, sp()))
    ref = taps["logmel"][:, 0]
    # fp32 FFT vs the reference's fp32 dense-DFT conv: bins far below the frame's peak are rounding
    # noise in BOTH; compare in dB where the bin is within 90 dB of the clip maximum, and loosely elsewhere
    d = (out.cpu() - ref).abs()
    strong = ref > (ref.max() - 90.0)
    assert float(d[strong].max()) < 0.02, float(d[strong].max())
    assert float(d.mean()) < 0.05
    out2 = torch.empty_like(out)
    _ffi.check(_ffi.lib().acx_logmel_bn0(ctx.handle, _ffi.ptr(wav), B, L, _ffi.ptr(out2), 1, sp()))
    d2 = (out2.cpu() - taps["bn0"][:, 0]).abs()
    assert float(d2[strong].max()) < 5e-3


def test_logmel_kernel_vs_torch_stft(ctx, golden_dir):
    """The frontend kernel against a path that shares no code with the oracle's restatement of torchlibrosa: torch.stft
    (scipy's periodic hann, center / reflect, hop 320) -> power -> mel matrix -> dB, in float64, on the reference's demo
    clip (convnext.py:179-200, 298-299).  Same dB criterion as test_logmel_kernel."""
    import scipy.signal
    from oracle import torchlibrosa_spec as tls
    pcm = np.load(os.path.join(golden_dir, "g1_demo.npz"))["pcm16"]
    wav = torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None, :]
    L = wav.shape[1]
    T = _ffi.num_frames(L)
    out = torch.empty(1, T, 224, device="cuda")
    wav_d = wav.cuda()
    _ffi.check(_ffi.lib().acx_logmel_bn0(ctx.handle, _ffi.ptr(wav_d), 1, L, _ffi.ptr(out), 0, sp()))
    win = torch.from_numpy(scipy.signal.get_window("hann", 1024, fftbins=True))
    st = torch.stft(wav.double(), 1024, 320, 1024, window=win, center=True, pad_mode="reflect", return_complex=True)
    power = (st.abs() ** 2).transpose(1, 2)                                       # (1, T, 513)
    assert power.shape[1] == T
    mel = power @ torch.from_numpy(tls.melW()).double()
    ref = 10.0 * torch.log10(mel.clamp_min(1e-10))
    d = (out.cpu().double() - ref).abs()
    strong = ref > (ref.max() - 90.0)
    assert float(d[strong].max()) < 0.02, float(d[strong].max())
    assert float(d.mean()) < 0.05


def test_stem_kernel(ctx, taps):
    x = taps["bn0"][:, 0].contiguous().cuda()          # (B,T,224)
    B, T, _ = x.shape
    ref = taps["ds0"]
    out = torch.empty(B, ref.shape[2], 56, 96, device="cuda")
    _ffi.check(_ffi.lib().acx_stem_ln(ctx.handle, _ffi.ptr(x), B, T, _ffi.ptr(out), sp()))
    assert maxdiff(nchw(out), ref) < LAYER_TOL


@pytest.mark.parametrize("B,T", [(1, 101), (3, 23), (1, 5), (5, 1001), (64, 311)])
def test_stem_kernel_shapes(ctx, synth_sd, B, T):
    """The stem walks output rows in PAIRS (two rows per workgroup step, two pixels per 32-lane group: stem.hip): odd row counts
    (1 x 27, 3 x 7 rows), a single pair, pairs that straddle two clips, time rows in the zero padding at both ends, and launches
    whose workgroups walk several pairs -- against Conv2d(1, 96, 4x4, stride 4, padding (4, 0)) + channels-first LayerNorm in
    float64 (convnext.py:688-691, :536-541)."""
    g = torch.Generator().manual_seed(100 * B + T)
    x = (torch.randn(B, T, 224, generator=g) * 3.0).cuda()
    H0 = (T + 8 - 4) // 4 + 1
    out = torch.full((B, H0, 56, 96), float("nan"), device="cuda")
    _ffi.check(_ffi.lib().acx_stem_ln(ctx.handle, _ffi.ptr(x), B, T, _ffi.ptr(out), sp()))
    w = synth_sd["downsample_layers.0.0.weight"].double()
    b = synth_sd["downsample_layers.0.0.bias"].double()
    lw = synth_sd["downsample_layers.0.1.weight"].double()
    lb = synth_sd["downsample_layers.0.1.bias"].double()
    y = torch.nn.functional.conv2d(x.cpu().double()[:, None], w, b, stride=4, padding=(4, 0))      # (B, 96, H0, 56)
    u = y.mean(1, keepdim=True)
    v = (y - u).pow(2).mean(1, keepdim=True)
    ref = (y - u) / torch.sqrt(v + 1e-6) * lw[None, :, None, None] + lb[None, :, None, None]
    assert ref.shape[2] == H0
    assert torch.isfinite(out).all()
    assert float((out.cpu().double().permute(0, 3, 1, 2) - ref).abs().max()) < LAYER_TOL


@pytest.mark.parametrize("s", [0, 1, 2, 3])
def test_dwconv_and_ln_stats(ctx, taps, synth_sd, s):
    x = nhwc(taps["ds%d" % s])
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(B * H * W, 2, device="cuda")
    _ffi.check(_ffi.lib().acx_dwconv7(ctx.handle, s, 0, _ffi.ptr(x), _ffi.ptr(y), _ffi.ptr(stats), B, H, W, sp()))
    assert maxdiff(nchw(y), taps["s%d.b0.dwconv" % s]) < LAYER_TOL
    # LayerNorm from the emitted statistics + the block's affine == the reference's F.layer_norm output
    w = synth_sd["stages.%d.0.norm.weight" % s].cuda()
    b = synth_sd["stages.%d.0.norm.bias" % s].cuda()
    ln = (y.view(-1, C) - stats[:, :1]) * stats[:, 1:] * w + b
    assert maxdiff(ln.view(B, H, W, C), taps["s%d.b0.ln" % s]) < LAYER_TOL


# The depthwise kernel streams the whole batch as ONE stacked image (clip, 3 zero rows, clip, ...) in row tiles of
# 8 / 8 / 16 / 32 rows: heights around the tile sizes, heights below the 7-row window, batches whose stacked height is
# not a multiple of anything, and a batch long enough for every workgroup to cross several clip boundaries.
@pytest.mark.parametrize("s,B,H", [(0, 1, 1), (0, 3, 5), (0, 2, 8), (0, 5, 13), (1, 1, 2), (1, 7, 9), (1, 2, 126),
                                   (2, 1, 3), (2, 9, 16), (2, 4, 17), (2, 33, 5), (3, 1, 1), (3, 2, 29), (3, 70, 31),
                                   (3, 3, 32), (3, 5, 33), (3, 6, 94)])
def test_dwconv_shapes_vs_torch(ctx, synth_sd, s, B, H):
    C, W = DIMS[s], 56 >> s
    g = torch.Generator().manual_seed(100 * s + 7 * B + H)
    x = torch.randn(B, H, W, C, generator=g).cuda()
    x[0, 0, 0, :] = 3.0                      # corners: catch a shifted or wrapped window
    x[-1, -1, -1, :] = -2.0
    y = torch.full_like(x, float("nan"))
    _ffi.check(_ffi.lib().acx_dwconv7(ctx.handle, s, 1, _ffi.ptr(x), _ffi.ptr(y), None, B, H, W, sp()))
    wt = synth_sd["stages.%d.1.dwconv.weight" % s].cuda()
    bias = synth_sd["stages.%d.1.dwconv.bias" % s].cuda()
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), padding=3, groups=C)
    assert bool(torch.isfinite(y).all())
    assert maxdiff(nchw(y), ref) < LAYER_TOL


@pytest.mark.parametrize("s", [0, 1, 2, 3])
def test_block(ctx, taps, s):
    x = nhwc(taps["ds%d" % s])
    B, H, W, C = x.shape
    need = ctypes.c_size_t()
    _ffi.check(_ffi.lib().acx_block_scratch_bytes(s, B, H, W, ctypes.byref(need)))
    scratch = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    _ffi.check(_ffi.lib().acx_block(ctx.handle, s, 0, _ffi.ptr(x), B, H, W, _ffi.ptr(scratch), need.value, sp()))
    d = maxdiff(nchw(x), taps["s%d.b0.out" % s])
    print("block stage %d: max abs diff vs reference tap %.3g" % (s, d))
    assert d < LAYER_TOL


@pytest.mark.parametrize("i", [1, 2, 3])
def test_downsample(ctx, taps, i):
    x = nhwc(taps["stage%d" % (i - 1)])
    B, H, W, C = x.shape
    ref = taps["ds%d" % i]
    out = torch.empty(B, H // 2, W // 2, DIMS[i], device="cuda")
    scratch = torch.empty_like(x)
    _ffi.check(_ffi.lib().acx_downsample(ctx.handle, i, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scratch), B, H, W, sp()))
    d = maxdiff(nchw(out), ref)
    print("downsample %d: max abs diff vs reference tap %.3g" % (i, d))
    assert d < LAYER_TOL


def test_pool_head(ctx, taps):
    x = nhwc(taps["stage3"])
    B, H3 = x.shape[0], x.shape[1]
    scene = torch.empty(B, 768, device="cuda")
    logits = torch.empty(B, 527, device="cuda")
    probs = torch.empty(B, 527, device="cuda")
    _ffi.check(_ffi.lib().acx_pool_head(ctx.handle, _ffi.ptr(x), B, H3, _ffi.ptr(scene), _ffi.ptr(logits),
                                        _ffi.ptr(probs), sp()))
    assert maxdiff(scene, taps["scene"]) < LAYER_TOL
    assert maxdiff(logits, taps["logits"]) < LAYER_TOL
    assert maxdiff(probs, taps["probs"]) < LAYER_TOL


def test_nhwc_to_nchw():
    x = torch.randn(3, 31, 7, 768, device="cuda")
    out = torch.empty(3, 768, 31, 7, device="cuda")
    _ffi.check(_ffi.lib().acx_nhwc_to_nchw(_ffi.ptr(x), _ffi.ptr(out), 3, 31, 7, 768, sp()))
    assert torch.equal(out, x.permute(0, 3, 1, 2))


def _check_outputs(model, wav, g, tol=E2E_TOL):
    wav = wav.cuda()
    out = model(wav)
    assert list(out.keys()) == ["clipwise_output", "clipwise_logits"]
    res = {
        "logits": maxdiff(out["clipwise_logits"], torch.from_numpy(g["logits"])),
        "probs": maxdiff(out["clipwise_output"], torch.from_numpy(g["probs"])),
        "scene": maxdiff(model.forward_scene_embeddings(wav), torch.from_numpy(g["scene"])),
        "frame": maxdiff(model.forward_frame_embeddings(wav), torch.from_numpy(g["frame"])),
    }
    print("max abs deviation vs reference:", res)
    return res


def test_e2e_golden_short_clips(model, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_taps.npz"))
    res = _check_outputs(model, torch.from_numpy(g["wav"]), g)
    parity_floor.check("parity/g2_taps/" + model.precision, res, E2E_TOL)


def test_e2e_golden_edge_signals(model, golden_dir):
    """1 s clips (odd H at every stage): noise, digital silence (the -100 dB clamp), full-scale square,
    a -80 dBFS sweep."""
    g = np.load(os.path.join(golden_dir, "g2_edge.npz"))
    res = _check_outputs(model, torch.from_numpy(g["wav"]), g)
    parity_floor.check("parity/g2_edge/" + model.precision, res, E2E_TOL)


def test_e2e_golden_demo_clip(model, golden_dir):
    """The reference's demo clip (audio_samples/f62-S-v2swA_200000_210000.wav, PCM16/32768), 10 s."""
    g = np.load(os.path.join(golden_dir, "g1_demo.npz"))
    wav = torch.from_numpy(g["pcm16"].astype(np.float32) / 32768.0)[None]
    out = model(wav.cuda())
    assert out["clipwise_logits"].shape == (1, 527) and out["clipwise_output"].shape == (1, 527)
    assert model.forward_scene_embeddings(wav.cuda()).shape == (1, 768)
    assert model.forward_frame_embeddings(wav.cuda()).shape == (1, 768, 31, 7)
    res = _check_outputs(model, wav, g)
    parity_floor.check("parity/g1_demo/" + model.precision, res, E2E_TOL)
    # same label set at the demo's 0.25 threshold (demo_convnext.py:87-88)
    ref_lbl = np.where(g["probs"][0] > 0.25)[0]
    got_lbl = np.where(out["clipwise_output"][0].cpu().numpy() > 0.25)[0]
    assert np.array_equal(ref_lbl, got_lbl)


def test_e2e_vs_oracle_random_batch(model, synth_sd):
    """Ragged batch (B=3) of 1.5 s noise clips against the CPU oracle on the same seeded input."""
    from oracle import ref_cpu
    wav = synth.synth_waveforms(3, 48000, seed=77)
    ref = ref_cpu.forward(synth_sd, wav)
    out = model(wav.cuda())
    res = {"logits": maxdiff(out["clipwise_logits"], ref["clipwise_logits"]),
           "probs": maxdiff(out["clipwise_output"], ref["clipwise_output"]),
           "frame": maxdiff(model.forward_frame_embeddings(wav.cuda()), ref_cpu.forward_frame_embeddings(synth_sd, wav))}
    print("ragged batch vs oracle:", res)
    parity_floor.check("parity/ragged_b3/" + model.precision, res, E2E_TOL)


def test_full_size_batch_properties(model, synth_sd):
    """BASELINE config 2 size (B=64, 10 s): clips are independent, so every clip of the batch must equal
    the same clip run alone (bit for bit: each GEMM row accumulates in the same order whatever the tile it
    lands in), and one clip is checked against the oracle."""
    from oracle import ref_cpu
    wav = synth.synth_waveforms(64, 320000, seed=1234).cuda()
    out = model(wav)["clipwise_logits"]
    assert out.shape == (64, 527) and bool(torch.isfinite(out).all())
    for b in (0, 37, 63):
        solo = model(wav[b:b + 1])["clipwise_logits"]
        assert torch.equal(solo[0], out[b]), (b, maxdiff(solo[0], out[b]))
    ref = ref_cpu.forward(synth_sd, wav[5:6].cpu())["clipwise_logits"]
    parity_floor.check("parity/bs64_clip5/" + model.precision, {"logits": maxdiff(out[5:6], ref)}, E2E_TOL)
    fr = model.forward_frame_embeddings(wav[:4])
    assert fr.shape == (4, 768, 31, 7)


def test_error_behaviour(model):
    with pytest.raises(RuntimeError, match="too short"):
        model(torch.zeros(1, 7359, device="cuda"))
    model(torch.zeros(1, 7360, device="cuda"))          # shortest legal clip
    with pytest.raises(RuntimeError, match="GPU only"):
        model(torch.zeros(1, 32000))
    model.train()
    try:
        with pytest.raises(RuntimeError, match="inference-only"):
            model(torch.zeros(1, 32000, device="cuda"))
    finally:
        model.eval()


def test_weight_reload_is_picked_up(synth_sd):
    m = convnext_tiny(after_stem_dim=[252, 56]).to("cuda").eval()
    wav = synth.synth_waveforms(1, 16000, seed=3).cuda()
    a = m(wav)["clipwise_logits"].clone()
    m.load_state_dict(synth_sd)
    b = m(wav)["clipwise_logits"]
    assert maxdiff(a, b) > 1e-2
    ref = convnext_tiny(after_stem_dim=[252, 56])
    ref.load_state_dict(synth_sd)
    c = ref.to("cuda").eval()(wav)["clipwise_logits"]
    assert torch.equal(b, c)


def test_workspace_growth_and_pruning(synth_sd):
    """Host-side scratch policy: one workspace per (device, stream), geometric growth, nothing retired unless a
    stream capture has seen it, at most _WS_MAX_STREAMS live workspaces."""
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
    m.load_state_dict(synth_sd)
    m = m.to("cuda").eval()
    wav = synth.synth_waveforms(1, 64000, seed=2).cuda()
    sizes = []
    for L in (16000, 17000, 18000, 30000, 64000):            # a length-sorted sweep: each clip a little longer
        m(wav[:, :L])
        assert len(m._ws) == 1 and not m._ws_retired
        sizes.append(next(iter(m._ws.values())).numel())
    torch.cuda.synchronize()
    grown = [b for a, b in zip(sizes, sizes[1:]) if b != a]
    assert len(grown) <= 3 and all(b >= a + a // 2 for a, b in zip(sizes, sizes[1:]) if b != a), sizes
    ref = m(wav[:, :16000])["clipwise_logits"].clone()
    streams = [torch.cuda.Stream() for _ in range(m._WS_MAX_STREAMS + 3)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            out = m(wav[:, :16000])["clipwise_logits"]
        st.synchronize()
        assert torch.equal(out, ref)
    assert len(m._ws) == m._WS_MAX_STREAMS and not m._ws_retired


def test_forward_is_graph_capturable(model):
    """Nothing in the launch path allocates or synchronises (include/acx.h): the whole forward can be captured
    into a hipGraph and replayed; replay == eager bit for bit."""
    wav = synth.synth_waveforms(2, 32000, seed=9).cuda()
    eager = model(wav)["clipwise_logits"].clone()
    torch.cuda.synchronize()          # the side stream below shares the model's workspace with this eager call
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model(wav)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = model(wav)["clipwise_logits"]
    torch.cuda.synchronize()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_split_batch_is_graph_capturable_and_exact(model):
    """B >= 16 runs as two halves on two streams inside acx_forward (event fork/join): still one capturable unit,
    and every clip equals the un-split result bit for bit."""
    wav = synth.synth_waveforms(17, 16000, seed=10).cuda()          # odd split 9 + 8
    whole = model(wav)
    parts = torch.cat([model(wav[:9])["clipwise_logits"], model(wav[9:])["clipwise_logits"]])
    assert torch.equal(whole["clipwise_logits"], parts)
    fr = model.forward_frame_embeddings(wav)
    assert torch.equal(fr[10], model.forward_frame_embeddings(wav[10:11])[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model(wav)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = model(wav)["clipwise_logits"]
    torch.cuda.synchronize()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, parts)


@pytest.mark.parametrize("precision", ["fp32", "fp32_split"])
def test_weight_scale_extremes(synth_sd, precision):
    """Per-layer power-of-two weight scaling of the split arithmetic (acx_finalize): an all-zero pwconv2, a pwconv1
    scaled down by 1e-6 and a pwconv2 scaled up by 64 must still track the oracle (the fp32 path takes the same
    test so that the tolerances are comparable)."""
    from oracle import ref_cpu
    sd = {k: v.clone() for k, v in synth_sd.items()}
    sd["stages.0.1.pwconv2.weight"].zero_()
    sd["stages.1.0.pwconv1.weight"].mul_(1e-6)
    sd["stages.2.3.pwconv2.weight"].mul_(64.0)
    sd["stages.2.3.gamma"].mul_(1.0 / 64.0)
    sd["downsample_layers.2.1.weight"].mul_(1e-3)
    sd["stages.3.1.pwconv1.weight"].mul_(1e-14)          # accumulator scale 2^75: the GELU argument must not overflow
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                      use_speed_perturb=False)
    m.load_state_dict(sd)
    m = m.to("cuda").eval().set_precision(precision)
    wav = synth.synth_waveforms(2, 40000, seed=21)
    ref = ref_cpu.forward(sd, wav)
    out = m(wav.cuda())
    d = maxdiff(out["clipwise_logits"], ref["clipwise_logits"])
    print("weight-scale extremes, %s: logits max abs diff %.3g" % (precision, d))
    assert d < E2E_TOL
    assert maxdiff(m.forward_frame_embeddings(wav.cuda()), ref_cpu.forward_frame_embeddings(sd, wav)) < E2E_TOL


@pytest.mark.parametrize("L", [96123, 960000])
def test_e2e_odd_and_long_clips(model, synth_sd, L):
    """MANIFEST.json shape cases: an odd length (T = 301, stage heights 76/38/19/9) and a 30 s clip (T = 3001):
    shapes as the reference produces them and values against the oracle."""
    from oracle import ref_cpu
    wav = synth.synth_waveforms(1, L, seed=400 + L % 7)
    ref = ref_cpu.forward(synth_sd, wav)
    out = model(wav.cuda())
    fr = model.forward_frame_embeddings(wav.cuda())
    h3, w3 = ref_cpu.out_hw(L)[3]
    assert out["clipwise_logits"].shape == (1, 527) and fr.shape == (1, 768, h3, w3)
    assert maxdiff(out["clipwise_logits"], ref["clipwise_logits"]) < E2E_TOL
    assert maxdiff(fr, ref_cpu.forward_frame_embeddings(synth_sd, wav)) < E2E_TOL


def test_default_capture_stream_at_split_batch_sizes(model):
    """`with torch.cuda.graph(g): model(x)` captures on torch's private capture stream, which nobody can warm up.
    A batch that would run as two sub-batches finds no fork / join set for that stream inside the capture: it runs un-split
    there (same bits, include/acx.h) instead of failing."""
    wav = synth.synth_waveforms(18, 16000, seed=12).cuda()
    eager = model(wav)["clipwise_logits"].clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(wav)["clipwise_logits"]
    torch.cuda.synchronize()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_failure_in_a_sub_batch_joins_the_forked_streams(model):
    """an error while queueing sub-batch 1 must not leave the side stream un-joined -- inside a stream
    capture an un-joined fork invalidates the capture (hipErrorStreamCaptureUnjoined at capture end), and the caller may free
    the workspace the side stream still uses.  The failure is injected by the library's test call acx_test_fail_sub (not an
    environment variable: ADVICE r04); the capture must END cleanly, and the model must work afterwards."""
    handle = model.native_context(torch.device("cuda", 0)).handle

    def fail_sub(sub):                    # per context: other contexts of the process are not affected
        return _ffi.lib().acx_test_fail_sub(handle, sub)
    wav = synth.synth_waveforms(17, 16000, seed=13).cuda()
    good = model(wav)["clipwise_logits"].clone()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):            # the stream gets its fork / join set here: the captures below DO split
        model(wav)
    torch.cuda.synchronize()
    try:
        for sub in (0, 1):
            assert fail_sub(sub) == 0
            with pytest.raises(_ffi.AcxError, match="acx_test_fail_sub"):        # eager
                model(wav)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                raised = False
                with torch.cuda.graph(g, stream=s):
                    try:
                        model(wav)
                    except _ffi.AcxError:
                        raised = True
                assert raised                                                    # ... and leaving the `with` did not raise: the fork was joined
            torch.cuda.synchronize()
    finally:
        fail_sub(-1)
    with torch.cuda.stream(s):
        again = model(wav)["clipwise_logits"]
    torch.cuda.synchronize()
    assert torch.equal(again, good)
