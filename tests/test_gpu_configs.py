"""GPU (`-m gpu`): BASELINE.json's configurations at THEIR sizes -- frame-embedding extraction at bs = 256 (configs[3]),
the per-rank shard of the 8-GPU bf16 run, bs = 64 x 10 s (configs[2]), an evaluation sweep over 2 048 clips in batches
of 256 (configs[4], world size 1 here; the sharding is covered on CPU by tests/test_parallel_cpu.py), the bs = 64 fp32
batch against the oracle on several clips and all three outputs (configs[1]) -- plus checkpoint files through the HIP
path and the log-mel kernel on the clamp cases.

At these sizes the oracle (seconds per clip on the CPU) checks a few clips; the rest is covered by a size-independent
property of the domain: clips are independent in eval mode, so any clip of a batch must equal the same clip run
alone, bit for bit."""
import os

import numpy as np
import pytest
import torch

from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch import evaluate as ev
from audioset_convnext_inf_amd.pytorch.convnext import ConvNeXt, convnext_tiny, load_checkpoint
from audioset_convnext_inf_amd.utils import utilities as ut
from audioset_convnext_inf_amd.utils.data_generator import ClipShard, evaluate_batches
import parity_floor

pytestmark = pytest.mark.gpu
E2E_TOL = 1e-3
L10 = 320000


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def make_model(sd, precision="fp32_split"):
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
    m.load_state_dict(sd)
    return m.to("cuda").eval().set_precision(precision)


@pytest.fixture(scope="module")
def model(synth_sd):
    return make_model(synth_sd)


def test_frame_embeddings_bs256(model, synth_sd):
    """configs[3]: (256, 768, 31, 7) from 256 x 10 s clips in one call."""
    from oracle import ref_cpu
    wav = synth.synth_waveforms(256, L10, seed=4321).cuda()
    fr = model.forward_frame_embeddings(wav)
    assert fr.shape == (256, 768, 31, 7) and bool(torch.isfinite(fr).all())
    for b in (0, 129, 255):
        assert torch.equal(model.forward_frame_embeddings(wav[b:b + 1])[0], fr[b]), b
    sc = model.forward_scene_embeddings(wav)
    assert sc.shape == (256, 768)
    pick = [7, 200]
    ref_fr = ref_cpu.forward_frame_embeddings(synth_sd, wav[pick].cpu())
    ref_sc = ref_cpu.forward_scene_embeddings(synth_sd, wav[pick].cpu())
    parity_floor.check("configs/frame_bs256/fp32_split", {"frame": maxdiff(fr[pick], ref_fr), "scene": maxdiff(sc[pick], ref_sc)}, E2E_TOL)


@pytest.mark.parametrize("precision", ["fp32", "fp32_split"])
def test_bs64_fp32_four_clips_all_outputs_vs_oracle(synth_sd, precision):
    """configs[1]: the bench workload itself; clips 3, 17, 40, 63 against the oracle on logits, probs, scene and frame."""
    from oracle import ref_cpu
    m = make_model(synth_sd, precision)
    wav = synth.synth_waveforms(64, L10, seed=1234).cuda()
    out = m(wav)
    sc = m.forward_scene_embeddings(wav)
    fr = m.forward_frame_embeddings(wav)
    pick = [3, 17, 40, 63]
    sub = wav[pick].cpu()
    taps = {}
    ref = ref_cpu.forward(synth_sd, sub, taps)
    res = {"logits": maxdiff(out["clipwise_logits"][pick], ref["clipwise_logits"]),
           "probs": maxdiff(out["clipwise_output"][pick], ref["clipwise_output"]),
           "scene": maxdiff(sc[pick], taps["scene"]),
           "frame": maxdiff(fr[pick], taps["stage3"])}
    print("bs=64 x 10 s, %s, 4 clips vs oracle:" % precision, res)
    parity_floor.check("configs/bs64_four_clips/" + precision, res, E2E_TOL)


@pytest.mark.parametrize("mode", ["bf16", "bf16a"])
def test_bf16_bs64_ten_seconds(synth_sd, mode):
    """configs[2], one rank's shard: 64 x 10 s in bf16 arithmetic ("bf16a": with the activations of stages 0-2 stored as bf16
    too) -- finite, clip-independent, and within the drift bf16 operands allow of the fp32-grade result (same bounds as
    tests/test_gpu_bf16.py uses at small sizes)."""
    m16 = make_model(synth_sd, mode)
    m32 = make_model(synth_sd, "fp32_split")
    wav = synth.synth_waveforms(64, L10, seed=99).cuda()
    o16 = m16(wav)
    assert bool(torch.isfinite(o16["clipwise_logits"]).all())
    for b in (0, 31, 63):
        assert torch.equal(m16(wav[b:b + 1])["clipwise_logits"][0], o16["clipwise_logits"][b]), b
    o32 = m32(wav)
    d = maxdiff(o16["clipwise_logits"], o32["clipwise_logits"])
    print("%s vs fp32_split at bs=64 x 10 s: logits max abs diff %.3g" % (mode, d))
    # contract: the drift bound of tests/test_gpu_bf16.py; regression bar: 3 x the drift recorded for this build's rounding points
    parity_floor.check("configs/bs64_drift/" + mode, {"logits": d}, 0.25, factor=3.0)
    agree = (o16["clipwise_output"] > 0.25) == (o32["clipwise_output"] > 0.25)
    assert float(agree.float().mean()) > 0.995
    fr = m16.forward_frame_embeddings(wav[:8])
    assert fr.shape == (8, 768, 31, 7) and bool(torch.isfinite(fr).all())


def test_eval_sweep_2048_clips(model, synth_sd):
    """configs[4] at world size 1: 2 048 synthetic int16 clips of 10 s, batches of 256 (int16 -> /32767 -> H2D -> model),
    mAP / AUC / d' as evaluate.py:44-58 computes them.  The sweep at batch 256 equals the sweep at batch 64 bit for
    bit, 8 clips match the oracle, and the metrics equal a recomputation from the gathered scores."""
    import time
    from oracle import ref_cpu
    n = 2048
    g = np.random.Generator(np.random.PCG64(11))
    wav = g.integers(-3277, 3277, size=(n, L10), dtype=np.int16)          # ~ -20 dBFS uniform noise
    tgt = g.random((n, 527)) < 0.05
    tgt[0], tgt[1] = True, False                                          # every class has both labels
    shard = ClipShard(wav, tgt)
    t0 = time.perf_counter()
    out = ev.forward(model, evaluate_batches(shard, batch_size=256), return_target=True)
    dt = time.perf_counter() - t0
    print("eval sweep: %d clips in %.2f s = %.0f clips/s including int16 -> float32 on the host and H2D" % (n, dt, n / dt))
    assert out["clipwise_output"].shape == (n, 527) and np.isfinite(out["clipwise_output"]).all()
    # the same sweep with the clips crossing PCIe as int16 and /32767 done on the GPU (what evaluate_sharded does): same bits
    ev.forward(model, evaluate_batches(shard, batch_size=256, device_cast=True))       # pinned buffers allocated
    t0 = time.perf_counter()
    fast = ev.forward(model, evaluate_batches(shard, batch_size=256, device_cast=True), return_target=True)
    dt = time.perf_counter() - t0
    print("eval sweep, int16 over PCIe + cast on the GPU: %d clips in %.2f s = %.0f clips/s" % (n, dt, n / dt))
    assert np.array_equal(fast["clipwise_output"], out["clipwise_output"]) and np.array_equal(fast["target"], out["target"])
    # against the same batches already resident in HBM (the feed must not be what limits a rank)
    res = torch.from_numpy(ut.int16_to_float32(wav[:256])).cuda()
    for _ in range(2):
        model(res)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 256):
        model(res)
    torch.cuda.synchronize()
    dt_res = time.perf_counter() - t0
    print("resident batches of 256: %.0f clips/s; the sweep with host staging and PCIe runs at %.2f of that" % (n / dt_res, dt_res / dt))
    # (8 batches: pipeline fill and drain are a tenth of such a sweep; the steady state -- bench.py `eval_sweep`, tools/lab/sweep_repeat.py --
    # is 0.95-0.99 on most boxes of the pool; round 4's pipeline: 0.73).  A wall-clock ratio on a shared 8-core host is a MEASUREMENT,
    # not a correctness gate: it is reported by bench.py; the bar is enforced only on request.
    if os.environ.get("ACX_TEST_TIMING") == "1":
        assert dt_res / dt > 0.75
    out64 = ev.forward(model, evaluate_batches(shard, batch_size=64))
    assert np.array_equal(out64["clipwise_output"], out["clipwise_output"])
    pick = [0, 255, 256, 1000, 1023, 1500, 2046, 2047]
    ref = ref_cpu.forward(synth_sd, torch.from_numpy(ut.int16_to_float32(wav[pick])))["clipwise_output"].numpy()
    parity_floor.check("configs/eval_sweep_8_clips/fp32_split", {"probs": float(np.abs(out["clipwise_output"][pick] - ref).max())}, E2E_TOL)
    stats = ev.Evaluator(model).evaluate(evaluate_batches(shard, batch_size=256))
    again = ev.calculate_statistics(out["target"], out["clipwise_output"])
    np.testing.assert_allclose(stats["average_precision"], again["average_precision"])
    np.testing.assert_allclose(stats["auc"], again["auc"])
    assert np.isfinite(stats["d_prime"]).all() and 0.0 < float(np.mean(stats["average_precision"])) < 1.0


def test_checkpoint_files_through_the_hip_path(tmp_path, synth_sd):
    """f3: `save_model` -> `ConvNeXt.from_pretrained(path)` (convnext.py:404-511) and `torch.save({"model": sd})` ->
    `load_checkpoint` (evaluate_convnext_on_audioset.py:36-38), each followed by the HIP forward against the oracle."""
    from oracle import ref_cpu
    from safetensors.torch import save_model
    src = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
    src.load_state_dict(synth_sd)
    st = str(tmp_path / "model.safetensors")
    save_model(src, st)
    pth = str(tmp_path / "convnext_tiny.pth")
    torch.save({"model": synth_sd}, pth)
    wav = synth.synth_waveforms(2, 48000, seed=31)
    ref = ref_cpu.forward(synth_sd, wav)
    ref_fr = ref_cpu.forward_frame_embeddings(synth_sd, wav)
    m1 = ConvNeXt.from_pretrained(st, map_location="cpu")
    assert sum(p.numel() for p in m1.parameters() if p.requires_grad) == 28222767
    m1 = m1.to("cuda").eval()
    m2 = load_checkpoint(convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                                       use_speed_perturb=False), pth).to("cuda").eval()
    for m in (m1, m2):
        out = m(wav.cuda())
        assert maxdiff(out["clipwise_logits"], ref["clipwise_logits"]) < E2E_TOL
        assert maxdiff(out["clipwise_output"], ref["clipwise_output"]) < E2E_TOL
        assert maxdiff(m.forward_frame_embeddings(wav.cuda()), ref_fr) < E2E_TOL
    assert torch.equal(m1(wav.cuda())["clipwise_logits"], m2(wav.cuda())["clipwise_logits"])


def test_refresh_after_a_data_edit(synth_sd):
    """Edits through `.data` do not bump autograd's version counters; `refresh()` makes the native weights follow."""
    m = make_model(synth_sd)
    wav = synth.synth_waveforms(1, 16000, seed=3).cuda()
    a = m(wav)["clipwise_logits"].clone()
    m.head_audioset.bias.data.add_(1.0)
    m.refresh()
    b = m(wav)["clipwise_logits"]
    assert maxdiff(b, a + 1.0) < 1e-5


def test_concurrent_forwards_on_two_streams(synth_sd):
    """Two forwards of the same module in flight on two torch streams (the reference module is re-entrant in eval):
    separate workspaces and fork/join events per stream, results equal the serial ones."""
    m = make_model(synth_sd, "fp32")            # batches of >= 16 clips also take the sub-batch split inside acx_forward
    w1 = synth.synth_waveforms(17, 32000, seed=41).cuda()
    w2 = synth.synth_waveforms(18, 32000, seed=42).cuda()
    r1 = m(w1)["clipwise_logits"].clone()
    r2 = m(w2)["clipwise_logits"].clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            o1 = m(w1)["clipwise_logits"]
        with torch.cuda.stream(s2):
            o2 = m(w2)["clipwise_logits"]
        torch.cuda.synchronize()
        assert torch.equal(o1, r1) and torch.equal(o2, r2)


def test_many_caller_streams_recycle_fork_join_sets(synth_sd):
    """A context called from ever-new caller streams (per-request streams, private capture streams): the library keeps at most 16
    fork / join sets, evicts the least recently used one nobody is queueing on and hands the evicted set to the next new stream
    (api.hip get_aux: the retired sets are a free pool) -- 40 streams in turn, then the first ones again, two of them
    at once: every result equals the serial one."""
    m = make_model(synth_sd, "fp32_split")
    w = synth.synth_waveforms(17, 32000, seed=43).cuda()          # 17 clips: two sub-batches, i.e. the fork / join path
    ref = m(w)["clipwise_logits"].clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(40)]
    outs = []
    for s in streams:
        with torch.cuda.stream(s):
            outs.append(m(w)["clipwise_logits"])
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    for a, b in ((0, 39), (1, 20), (5, 6)):                        # evicted long ago, evicted recently, both evicted
        with torch.cuda.stream(streams[a]):
            oa = m(w)["clipwise_logits"]
        with torch.cuda.stream(streams[b]):
            ob = m(w)["clipwise_logits"]
        torch.cuda.synchronize()
        assert torch.equal(oa, ref) and torch.equal(ob, ref)


@pytest.mark.parametrize("precision", ["fp32_split", "bf16a"])
def test_sub_batch_split_is_invisible(synth_sd, precision, monkeypatch):
    """acx_forward runs a batch as sub-batches on side streams (acx_sub_batches; default 2 from 16 clips up): whatever the
    number of sub-batches -- also an uneven 3-way split of 37 clips -- every clip's result is the one-stream result bit for bit."""
    wav = synth.synth_waveforms(37, 48000, seed=123).cuda()
    outs = {}
    for ways in ("1", "2", "3", "4"):
        monkeypatch.setenv("ACX_SPLIT_WAYS", ways)        # read at acx_create: a fresh module = a fresh context
        m = make_model(synth_sd, precision)
        ctx = m.native_context(wav.device)
        assert ctx.sub_batches(37) == int(ways) and ctx.sub_batches(15) == 1 and ctx.sub_batches(16) == min(int(ways), 2)
        outs[ways] = (m(wav)["clipwise_logits"].clone(), m.forward_frame_embeddings(wav).clone())
        torch.cuda.synchronize()
        del m
    for ways in ("2", "3", "4"):
        assert torch.equal(outs[ways][0], outs["1"][0]) and torch.equal(outs[ways][1], outs["1"][1]), ways
    monkeypatch.delenv("ACX_SPLIT_WAYS")
    m = make_model(synth_sd, precision)
    assert m.native_context(wav.device).sub_batches(64) == 2


@pytest.mark.parametrize("precision", ["fp32_split", "bf16", "bf16a"])
def test_forward_beside_foreign_work_on_another_stream(synth_sd, precision):
    """The round-1 defect: while a dense 16-bit-MFMA kernel ran, FFT-type kernels of OTHER streams (rocFFT included)
    returned wrong results.  Every such kernel now launches CU-exclusive workgroups; here a forward of
    the 16-bit arithmetics runs beside torch work on a side stream -- elementwise, softmax (LDS reductions) and rocFFT --
    and both sides must reproduce their serial results bit for bit."""
    m = make_model(synth_sd, precision)
    wav = synth.synth_waveforms(32, 160000, seed=77).cuda()
    ref_out = m(wav)["clipwise_logits"].clone()
    v = torch.randn(4096, 4096, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))

    def victims():
        a = torch.sin(v) * 2 + 1
        b = torch.softmax(v, dim=1)
        c = torch.fft.rfft(v[:2048, :1024], dim=1).abs()
        return a, b, c

    ref_v = victims()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    wrong_v, wrong_f = 0, 0
    for _ in range(10):
        torch.cuda.synchronize()
        out = m(wav)["clipwise_logits"]           # ~5 ms of kernels queued on the current stream
        with torch.cuda.stream(side):
            got = victims()
        torch.cuda.synchronize()
        wrong_v += sum(int(not torch.equal(g, r)) for g, r in zip(got, ref_v))
        wrong_f += int(not torch.equal(out, ref_out))
    assert wrong_v == 0, "foreign kernels disturbed in %d of 30 results" % wrong_v
    assert wrong_f == 0, "forward disturbed in %d of 10 runs" % wrong_f


def test_logmel_kernel_clamp_cases(model, synth_sd):
    """Kernel level (acx_logmel_bn0): digital silence sits exactly on the -100 dB clamp (10 log10 max(mel, 1e-10),
    convnext.py:190-200) and bn0 maps the clamp value as the reference does; a -80 dBFS tone matches the reference's
    dense-DFT log-mel on every bin that carries signal and stays at the floor elsewhere."""
    from oracle import ref_cpu
    ctx = model.native_context(torch.device("cuda", 0))
    L = 32000
    t = torch.arange(L, dtype=torch.float64) / 32000.0
    tone = (1e-4 * torch.sin(2 * np.pi * 1000.0 * t)).to(torch.float32)
    wav = torch.stack([torch.zeros(L), tone])
    B, T = 2, _ffi.num_frames(L)
    sp = _ffi.stream_ptr(torch.device("cuda", 0))
    raw = torch.empty(B, T, 224, device="cuda")
    bn = torch.empty(B, T, 224, device="cuda")
    _ffi.check(_ffi.lib().acx_logmel_bn0(ctx.handle, _ffi.ptr(wav.cuda()), B, L, _ffi.ptr(raw), 0, sp))
    _ffi.check(_ffi.lib().acx_logmel_bn0(ctx.handle, _ffi.ptr(wav.cuda()), B, L, _ffi.ptr(bn), 1, sp))
    taps = {}
    ref_cpu.frontend(synth_sd, wav, taps)
    ref_raw, ref_bn = taps["logmel"][:, 0], taps["bn0"][:, 0]
    # silence: the clamp, exactly as the reference's fp32 evaluates it
    assert float(ref_raw[0].max()) == float(ref_raw[0].min())
    assert maxdiff(raw[0], ref_raw[0]) < 1e-4
    assert maxdiff(bn[0], ref_bn[0]) < 1e-4
    # -80 dBFS tone: bins within 40 dB of the frame peak agree to 0.02 dB; no bin rises above the reference by more
    # than the fp32 noise floor of the two DFT formulations allows
    strong = ref_raw[1] > (ref_raw[1].max() - 40.0)
    assert int(strong.sum()) > 100
    assert float((raw[1].cpu() - ref_raw[1]).abs()[strong].max()) < 0.02
    assert float((bn[1].cpu() - ref_bn[1]).abs()[strong].max()) < 5e-3
    assert float(raw[1].min()) >= -100.0 - 1e-4


@pytest.mark.parametrize("precision", ["fp32_split", "bf16a"])
@pytest.mark.parametrize("B,L", [(1, 320000), (3, 96123), (20, 48000)])
def test_small_launch_tile_shapes_are_invisible(synth_sd, B, L, precision, monkeypatch):
    """Small launches use narrower tiles so that they spread over more CUs: the fused MLP of stages 1-2 runs 64-pixel tiles (one
    16-pixel block per wave) instead of 128 (mlp_fused_wide.hip, ACX_WIDE_NPB = 1 | 2), the split GEMM of stage 3 and of the
    downsample convs 64- or 128-row tiles instead of 256 (gemm_split.hip, ACX_GEMM_MI = 1 | 2 | 4) -- whenever all the narrower
    tiles find a CU at once.  The arithmetic of an output element does not depend on the tile shape: forcing any of them (the
    variables are re-read by acx_tuning_refresh) must give the same bits as the default choice.  (bf16 arithmetics: the GEMM switch
    selects 128- against 256-row tiles of gemm_bf16_kernel; the fused bf16 kernels have one tile shape.)"""
    m = make_model(synth_sd, precision)
    wav = synth.synth_waveforms(B, L, seed=900 + B).cuda()

    def run():
        out = (m(wav)["clipwise_logits"].clone(), m.forward_frame_embeddings(wav).clone())
        torch.cuda.synchronize()
        return out

    from audioset_convnext_inf_amd import _ffi
    refresh = _ffi.lib().acx_tuning_refresh
    for var in ("ACX_WIDE_NPB", "ACX_GEMM_MI", "ACX_DW_STREAM"):
        monkeypatch.delenv(var, raising=False)
    refresh()
    ref = run()
    try:
        for var, values in (("ACX_WIDE_NPB", ("1", "2")), ("ACX_GEMM_MI", ("1", "2", "4")), ("ACX_DW_STREAM", ("0", "1"))):
            for v in values:
                monkeypatch.setenv(var, v)
                refresh()
                got = run()
                assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (var, v)
            monkeypatch.delenv(var)
    finally:
        for var in ("ACX_WIDE_NPB", "ACX_GEMM_MI", "ACX_DW_STREAM"):
            monkeypatch.delenv(var, raising=False)
        refresh()


@pytest.mark.parametrize("B,L", [(24, 160000), (3, 96123)])
def test_persistent_fused_mlp_is_invisible(synth_sd, B, L, monkeypatch):
    """Stage 1's fused MLP (C = 192, fp32_split) runs as a persistent kernel once a CU has more than one tile to walk
    (mlp_fused_wide.hip, PERS): the next tile's rows are requested two segments ahead, the weight ring runs on across tiles and a
    tile's stores stay in flight under the next one -- all of it resting on hand-counted s_waitcnt vmcnt(N).  24 clips of 5 s give
    every workgroup several tiles and a last tile that is only partly inside M; 3 clips of 3 s force the form onto a launch whose
    workgroups have one tile each.  ACX_WIDE_PERSIST = 0 | 1 selects the form (ACX_WIDE_NPB = 2 keeps the 128-pixel tile that the
    persistent form is written for): same arithmetic, same bits, also on a second stream and under two sub-batches."""
    wav = synth.synth_waveforms(B, L, seed=1200 + B).cuda()

    def run(m):
        out = (m(wav)["clipwise_logits"].clone(), m.forward_frame_embeddings(wav).clone())
        torch.cuda.synchronize()
        return out

    from audioset_convnext_inf_amd import _ffi
    refresh = _ffi.lib().acx_tuning_refresh
    names = ("ACX_WIDE_NPB", "ACX_WIDE_PERSIST", "ACX_SPLIT_WAYS")
    for var in names:
        monkeypatch.delenv(var, raising=False)
    try:
        monkeypatch.setenv("ACX_WIDE_NPB", "2")
        monkeypatch.setenv("ACX_WIDE_PERSIST", "0")
        refresh()
        ref = run(make_model(synth_sd, "fp32_split"))
        monkeypatch.setenv("ACX_WIDE_PERSIST", "1")
        refresh()
        for ways in (None, "1", "3"):        # (read when the context is created)
            if ways is None:
                monkeypatch.delenv("ACX_SPLIT_WAYS", raising=False)
            else:
                monkeypatch.setenv("ACX_SPLIT_WAYS", ways)
            m = make_model(synth_sd, "fp32_split")
            for rep in range(3):
                got = run(m)
                assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (ways, rep)
    finally:
        for var in names:
            monkeypatch.delenv(var, raising=False)
        refresh()
