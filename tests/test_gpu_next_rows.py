"""GPU (`-m gpu`): the "next" rows end to end on the HIP path -- evaluation harness, variable-length
extraction, the demo script -- against the CPU oracle on the same seeded inputs."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch import evaluate as ev
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
from audioset_convnext_inf_amd.pytorch.extract_embeddings import extract
from audioset_convnext_inf_amd.utils import utilities as ut
from audioset_convnext_inf_amd.utils.data_generator import ClipShard, evaluate_batches

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(synth_sd):
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                      use_speed_perturb=False)
    m.load_state_dict(synth_sd)
    return m.to("cuda").eval()


def test_evaluator_matches_oracle(model, synth_sd):
    from oracle import ref_cpu
    rs = np.random.RandomState(5)
    n, L = 23, 32000                                   # 3 batches of 10, the last one short
    wav = (rs.standard_normal((n, L)) * 0.1 * 32767).astype(np.int16)
    tgt = rs.uniform(size=(n, 527)) < 0.3
    tgt[0], tgt[1] = True, False
    shard = ClipShard(wav, tgt)
    out = ev.forward(model, evaluate_batches(shard, batch_size=10), return_target=True)
    assert out["clipwise_output"].shape == (n, 527) and out["target"].shape == (n, 527)
    ref = ref_cpu.forward(synth_sd, torch.from_numpy(ut.int16_to_float32(wav)))["clipwise_output"].numpy()
    assert np.abs(out["clipwise_output"] - ref).max() < 1e-3          # BASELINE tolerance on probs
    stats = ev.Evaluator(model).evaluate(evaluate_batches(shard, batch_size=10))
    ref_stats = ev.calculate_statistics(tgt.astype(np.float32), ref)
    assert abs(np.mean(stats["average_precision"]) - np.mean(ref_stats["average_precision"])) < 2e-3
    assert abs(np.mean(stats["auc"]) - np.mean(ref_stats["auc"])) < 2e-3
    one = ev.evaluate_sharded(model, shard, batch_size=10)            # world_size 1 path of the sharded sweep
    np.testing.assert_allclose(one["average_precision"], stats["average_precision"])


def test_variable_length_extraction(model, synth_sd):
    from oracle import ref_cpu
    lengths = [7360, 9600, 7360, 32000, 9600]
    clips = [synth.synth_waveforms(1, n, seed=100 + i)[0] for i, n in enumerate(lengths)]
    got = extract(model, clips, "logits")
    frames = extract(model, clips, "frame")
    for i, (c, n) in enumerate(zip(clips, lengths)):
        solo = model(c[None].cuda())["clipwise_logits"][0].cpu()
        assert torch.equal(got[i], solo)                               # bucketing does not change a clip's result
        assert frames[i].shape == (768, ref_cpu.out_hw(n)[3][0], 7)
    ref = ref_cpu.forward(synth_sd, clips[3][None])["clipwise_logits"][0]
    assert float((got[3] - ref).abs().max()) < 1e-3


def test_demo_script(tmp_path):
    rs = np.random.RandomState(0)
    wav = str(tmp_path / "clip.wav")
    ut.write_wav_pcm16(wav, (rs.standard_normal(48000) * 0.1).astype(np.float32), 32000, list_chunk=True)
    labels = tmp_path / "labels.csv"
    labels.write_text("index,mid,display_name\n" + "".join('%d,/m/%04d,"label %d"\n' % (i, i, i) for i in range(527)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo_convnext.py"), "--synthetic-weights", "--wav", wav,
                        "--labels", str(labels)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "# params: 28222767" in out                                  # README.md:49
    assert "logits size: torch.Size([1, 527])" in out and "probs size: torch.Size([1, 527])" in out
    assert "Scene embedding, shape: torch.Size([1, 768])" in out
    assert "Frame-level embeddings, shape: torch.Size([1, 768, 31, 7])" in out   # padded to 10 s
    assert "Padding waveform" in out and "label " in out


def test_demo_script_probabilities_match_the_oracle(tmp_path):
    """not only the shapes -- the probabilities demo_convnext.py prints for the labels above its
    threshold are the oracle's for the same clip and weights (printed to 3 decimals)."""
    import re
    from oracle import ref_cpu
    rs = np.random.RandomState(3)
    t = np.arange(5 * 32000) / 32000.0
    clip = (0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.05 * rs.standard_normal(t.size)).astype(np.float32)
    wav = str(tmp_path / "tone.wav")
    ut.write_wav_pcm16(wav, clip, 32000, list_chunk=False)
    labels = tmp_path / "labels.csv"
    labels.write_text("index,mid,display_name\n" + "".join('%d,/m/%04d,"label %d"\n' % (i, i, i) for i in range(527)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo_convnext.py"), "--synthetic-weights", "--wav", wav,
                        "--labels", str(labels), "--threshold", "0.5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    printed = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"^label (\d+): ([0-9.]+)$", r.stdout, flags=re.M)}
    assert len(printed) >= 5, r.stdout[-1500:]
    pcm, sr = ut.read_wav_pcm16(wav)
    x = ut.prepare_clip(torch.from_numpy(pcm[:1]), sr, 32000, 10)
    probs = ref_cpu.forward(synth.synth_state_dict(0), x)["clipwise_output"][0]
    above = set(int(i) for i in np.where(probs.numpy() > 0.5)[0])
    # the same label set up to probabilities within rounding distance of the threshold, the same values to 3 decimals (+ 1e-3 parity)
    edge = set(int(i) for i in np.where(np.abs(probs.numpy() - 0.5) < 2e-3)[0])
    assert set(printed) - edge == above - edge
    for i, p in printed.items():
        assert abs(p - float(probs[i])) < 1.6e-3, (i, p, float(probs[i]))


def test_evaluate_script_synthetic_sweep(model, synth_sd):
    """evaluate_convnext_on_audioset.py itself (not only its callees): `--synthetic 300` prints the three
    `Validate <set> <metric>` lines of the reference (evaluate_convnext_on_audioset.py:77-85); the numbers equal the statistics of
    the same shard scored in this process, whose probabilities in turn match the oracle on a few clips."""
    import importlib.util
    import re
    from oracle import ref_cpu
    from audioset_convnext_inf_amd.pytorch.evaluate import calculate_statistics
    script = os.path.join(ROOT, "evaluate_convnext_on_audioset.py")
    r = subprocess.run([sys.executable, script, "--synthetic", "300", "--batch_size", "64"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = {k: float(v) for k, v in re.findall(r"^Validate synthetic (mAP|AUC|d-prime): ([0-9.\-]+)$", r.stdout, flags=re.M)}
    assert set(got) == {"mAP", "AUC", "d-prime"}, r.stdout[-1500:]
    assert "total_params 28222767" in r.stdout and "300 clips in" in r.stdout
    spec = importlib.util.spec_from_file_location("eval_script", script)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    shard = mod.synthetic_shard(300)
    scores = []
    for i in range(0, 300, 60):
        w = torch.from_numpy((shard.waveforms[i:i + 60] / 32767.0).astype(np.float32)).cuda()
        scores.append(model(w)["clipwise_output"].cpu().numpy())
    scores = np.concatenate(scores)
    st = calculate_statistics(np.asarray(shard.targets[:300], dtype=np.float32), scores)
    assert abs(got["mAP"] - float(np.mean(st["average_precision"]))) < 2e-3
    assert abs(got["AUC"] - float(np.mean(st["auc"]))) < 2e-3
    assert abs(got["d-prime"] - float(np.mean(st["d_prime"]))) < 5e-3
    w4 = torch.from_numpy((shard.waveforms[:3] / 32767.0).astype(np.float32))
    ref = ref_cpu.forward(synth_sd, w4)["clipwise_output"]
    assert float((torch.from_numpy(scores[:3]) - ref).abs().max()) < 1e-3


def test_pcm16_to_f32_is_the_reference_arithmetic():
    """acx_pcm16_to_f32 (include/acx.h) against utilities.int16_to_float32 -- (x / 32767.0).astype(float32), the reference's
    utils/utilities.py:226-227 -- on every int16 value, bit for bit, and on a buffer whose length is not a multiple of 8."""
    import ctypes
    import numpy as np
    import torch
    from audioset_convnext_inf_amd import _ffi
    from audioset_convnext_inf_amd.utils.utilities import int16_to_float32
    allv = np.arange(-32768, 32768, dtype=np.int16)
    x = np.concatenate([allv, allv[::-1], allv[:12345]])
    assert x.size % 8 != 0
    dev = torch.from_numpy(x).cuda()
    out = torch.empty(x.size, dtype=torch.float32, device="cuda")
    _ffi.check(_ffi.lib().acx_pcm16_to_f32(_ffi.ptr(dev), _ffi.ptr(out), x.size, _ffi.stream_ptr(dev.device)))
    torch.cuda.synchronize()
    want = int16_to_float32(x)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # through the evaluation harness' own helper (what pytorch/evaluate.py::forward calls)
    from audioset_convnext_inf_amd.pytorch.evaluate import pcm16_to_float32
    got = pcm16_to_float32(dev[:3 * 47805].view(3, -1)[:, :40000].contiguous())
    assert np.array_equal(got.cpu().numpy(), int16_to_float32(x[:3 * 47805].reshape(3, -1)[:, :40000]))
    rc = _ffi.lib().acx_pcm16_to_f32(ctypes.c_void_p(dev.data_ptr() + 2), _ffi.ptr(out), 8, _ffi.stream_ptr(dev.device))
    assert rc != 0 and b"aligned" in _ffi.lib().acx_last_error()
