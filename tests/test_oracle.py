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


def test_oracle_g5_frontend_probes(synth_sd, golden_dir):
    """The frontend probe family (tests/golden/make_frontend_goldens.py): oracle against the reference class's outputs.  Bit-equal in
    the build container; another host's BLAS may sum the STFT in another order, which on these probes moves the outputs by up to the
    recorded `reference_stft_rounding_sensitivity` (tests/golden/frontend_self_noise.py) -- the bar is twice that, at least 1e-5."""
    import hashlib
    g = np.load(os.path.join(golden_dir, "g5_frontend.npz"))
    sens = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["reference_stft_rounding_sensitivity"]
    for name, L in synth.FRONTEND_PROBES:
        if L > 32000 and name != "chirp_10s":
            continue                                            # one 10 s probe is enough on the CPU
        wav = synth.frontend_probe(name)
        assert hashlib.sha256(wav.numpy().tobytes()).digest() == bytes(g[name + "/sha256"])
        o = ref_cpu.forward(synth_sd, wav)
        fr = ref_cpu.forward_frame_embeddings(synth_sd, wav)
        for key, got in (("logits", o["clipwise_logits"]), ("frame", fr)):
            d = float((got.double() - torch.from_numpy(g[name + "/" + key]).double()).abs().max())
            assert d <= max(1e-5, 2.0 * sens[name][key]), (name, key, d)


def test_mel_bank_properties_independent_of_the_product_tables():
    """What librosa 0.8.1's `filters.mel(sr=32000, n_fft=1024, n_mels=224, fmin=50, fmax=14000)` must look like, checked on the
    matrix the PRODUCT ships (frontend_tables.mel_matrix) with arithmetic written here and nowhere else (the
    library itself is absent and no checkpoint is at hand to pin the table bit for bit):
      * Slaney scale: 226 points equally spaced in mel between mel(50 Hz) and mel(14 kHz), linear (200/3 Hz per mel) below the
        1 kHz knee, logarithmic (27 mel per factor 6.4) above -- every filter's centroid sits at the mean of its three points;
      * Slaney area normalisation: every triangle integrates to 1 over frequency;
      * triangles: non-negative, one contiguous band each, rising then falling, neighbours overlapping half-way."""
    m = ft.mel_matrix().astype(np.float64)                    # (513, 224)
    assert m.shape == (513, 224) and (m >= 0).all()
    bin_hz = 32000.0 / 1024.0
    f = np.arange(513) * bin_hz

    def mel(hz):
        return hz / (200.0 / 3.0) if hz < 1000.0 else 15.0 + np.log(hz / 1000.0) / (np.log(6.4) / 27.0)

    def hz(ml):
        return ml * (200.0 / 3.0) if ml < 15.0 else 1000.0 * np.exp((ml - 15.0) * (np.log(6.4) / 27.0))

    lo, hi = mel(50.0), mel(14000.0)
    pts = np.array([hz(lo + (hi - lo) * i / 225.0) for i in range(226)])
    assert abs(pts[0] - 50.0) < 1e-9 and abs(pts[-1] - 14000.0) < 1e-6
    knee = int(np.searchsorted(pts, 1000.0))
    assert 50 < knee < 70                                                        # 57 points lie below 1 kHz
    assert np.allclose(np.diff(pts[:knee]), np.diff(pts[:knee])[0], rtol=1e-9)   # equal steps in Hz below the knee
    assert np.allclose(pts[knee + 1:] / pts[knee:-1], pts[knee + 1] / pts[knee], rtol=1e-9)     # equal ratios above it
    n_wide = 0
    for c in range(224):
        w = m[:, c]
        nz = np.nonzero(w)[0]
        a, ctr, b = pts[c], pts[c + 1], pts[c + 2]
        # support: the bins strictly inside (a, b)
        inside = np.nonzero((f > a) & (f < b))[0]
        assert len(nz) >= 1 and nz[0] >= inside[0] and nz[-1] <= inside[-1] and len(nz) >= len(inside) - 0, (c, nz, inside)
        # the exact triangle, sampled: height 2 / (b - a) at ctr
        tri = np.maximum(0.0, np.minimum((f - a) / (ctr - a), (b - f) / (b - ctr))) * (2.0 / (b - a))
        assert np.abs(w - tri).max() <= 2e-7 * tri.max() + 1e-12, (c, np.abs(w - tri).max())
        # area 1 and centroid at the mean of the three points (a SAMPLED triangle: checked where it spans >= 10 bins)
        if len(nz) >= 10:
            n_wide += 1
            assert abs(w.sum() * bin_hz - 1.0) < 0.05, (c, w.sum() * bin_hz)
            assert abs((w * f).sum() / w.sum() - (a + ctr + b) / 3.0) < 0.5 * bin_hz
    assert n_wide >= 20
