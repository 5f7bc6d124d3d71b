"""CPU: the oracle (oracle/ref_cpu.py) against the golden vectors generated from the
reference model class (tests/golden/make_goldens.py), plus the frontend-table restatements."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu, torchlibrosa_spec as tls
from audioset_convnext_inf_amd import frontend_tables as ft, synth

TOL = 2e-5   # oracle == reference op for op; slack only for thread-count dependent reductions


def test_manifest_pins(synth_sd, golden_dir):
    m = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))
    assert m["trainable_params"] == 28222767            # README.md:49, config.yaml:5
    assert m["state_dict_keys"] == 190
    assert synth.state_dict_digest(synth_sd) == m["weights_sha256"]
    for tag, dev in m["oracle_vs_reference"].items():
        assert max(dev.values()) <= 1e-5, (tag, dev)
    assert m["shapes"]["g1_demo"]["frame"] == [1, 768, 31, 7]      # README.md:61
    assert m["shapes"]["g3_lengths"]["7359"].startswith("RuntimeError")


def test_trainable_param_count(synth_sd):
    frozen = ("spectrogram_extractor", "logmel_extractor", "running_", "num_batches")
    n = sum(v.numel() for k, v in synth_sd.items() if not any(f in k for f in frozen))
    assert n == 28222767


def test_oracle_g2_taps(synth_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_taps.npz"))
    wav = torch.from_numpy(g["wav"])
    taps = {}
    out = ref_cpu.forward(synth_sd, wav, taps=taps)
    taps["logits"], taps["probs"] = out["clipwise_logits"], out["clipwise_output"]
    taps["frame"] = ref_cpu.forward_frame_embeddings(synth_sd, wav)
    checked = 0
    for k in g.files:
        if k in taps:
            np.testing.assert_allclose(taps[k].numpy(), g[k], atol=TOL, rtol=0, err_msg=k)
            checked += 1
    assert checked >= 25


def test_oracle_g2_edge(synth_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_edge.npz"))
    wav = torch.from_numpy(g["wav"])
    out = ref_cpu.forward(synth_sd, wav)
    np.testing.assert_allclose(out["clipwise_logits"].numpy(), g["logits"], atol=TOL, rtol=0)
    np.testing.assert_allclose(out["clipwise_output"].numpy(), g["probs"], atol=TOL, rtol=0)
    np.testing.assert_allclose(ref_cpu.forward_scene_embeddings(synth_sd, wav).numpy(), g["scene"], atol=TOL, rtol=0)
    np.testing.assert_allclose(ref_cpu.forward_frame_embeddings(synth_sd, wav).numpy(), g["frame"], atol=TOL, rtol=0)
    # digital silence sits exactly on the amin clamp: log-mel == -100 dB everywhere
    lm = ref_cpu.logmel(synth_sd, ref_cpu.spectrogram(synth_sd, wav[1:2]))
    assert float(lm.max()) == -100.0 and float(lm.min()) == -100.0


def test_oracle_g1_demo(synth_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_demo.npz"))
    wav = torch.from_numpy(g["pcm16"].astype(np.float32) / 32768.0)[None]
    out = ref_cpu.forward(synth_sd, wav)
    assert out["clipwise_logits"].shape == (1, 527)
    np.testing.assert_allclose(out["clipwise_logits"].numpy(), g["logits"], atol=TOL, rtol=0)
    np.testing.assert_allclose(ref_cpu.forward_frame_embeddings(synth_sd, wav).numpy(), g["frame"], atol=TOL, rtol=0)


def test_oracle_g4_stress(golden_dir):
    """The oracle against the reference class under the trained-like stress weights (tests/golden/make_stress_golden.py)."""
    g = np.load(os.path.join(golden_dir, "g4_stress.npz"))
    m = json.load(open(os.path.join(golden_dir, "MANIFEST_stress.json")))
    sd = synth.stress_state_dict(0)
    assert synth.state_dict_digest(sd) == m["weights_sha256"]
    wav = torch.from_numpy(g["wav"])
    o = ref_cpu.forward(sd, wav)
    got = {"logits": o["clipwise_logits"], "probs": o["clipwise_output"],
           "scene": ref_cpu.forward_scene_embeddings(sd, wav), "frame": ref_cpu.forward_frame_embeddings(sd, wav)}
    for k, v in got.items():
        d = float((v.double() - torch.from_numpy(g[k]).double()).abs().max())
        assert d <= 1e-5 * max(1.0, m["max_abs_reference"][k]), (k, d)


def test_shape_contract(golden_dir):
    m = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["shapes"]["g3_lengths"]
    for L in (7360, 96123, 320000, 960000):
        assert list(ref_cpu.out_hw(L)[3]) == m[str(L)]["frame"][2:]
    assert ref_cpu.out_hw(7359)[3][0] == 0


def test_frontend_tables_agree():
    """Product tables (closed form) vs the oracle's step-by-step library restatement."""
    r, i = ft.stft_weights()
    r2, i2 = tls.stft_conv_weights()
    assert np.abs(r - r2).max() <= 6e-8 and np.abs(i - i2).max() <= 6e-8
    np.testing.assert_array_equal(ft.mel_matrix(), tls.melW())
    h = tls.hann_periodic()
    assert h[0] == 0.0 and abs(h[512] - 1.0) < 1e-15
    m = tls.melW()
    assert m.shape == (513, 224) and (m >= 0).all()
    # every filter is a single contiguous band inside [50, 14000] Hz
    for c in range(224):
        nz = np.nonzero(m[:, c])[0]
        assert len(nz) >= 1 and nz[-1] - nz[0] + 1 == len(nz)
        assert nz[0] * 31.25 >= 50 and nz[-1] * 31.25 <= 14000
