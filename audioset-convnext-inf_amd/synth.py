"""Seeded synthetic weights and inputs.

No checkpoint ships with the reference (weights live on Zenodo / HF, README.md:24-30) and
there is no network, so tests, goldens and the benchmark all derive the *same* 190-entry
state_dict from this recipe (numpy RandomState = frozen legacy stream).  The recipe is
non-degenerate on purpose: the reference initialises layer-scale gamma to 1e-6
(convnext.py:56,67-71), which would make every block an identity and hide kernel errors.
"""
import hashlib
from collections import OrderedDict

import numpy as np
import torch

from . import frontend_tables as ft

DEPTHS = (3, 3, 9, 3)
DIMS = (96, 192, 384, 768)
NUM_CLASSES = 527


def state_dict_spec():
    """Ordered (key, shape, dtype) list == reference `ConvNeXt.state_dict()` for convnext_tiny
    with the [252,56] stem (convnext.py:641-708)."""
    spec = [
        ("spectrogram_extractor.stft.conv_real.weight", (ft.N_BINS, 1, ft.N_FFT), torch.float32),
        ("spectrogram_extractor.stft.conv_imag.weight", (ft.N_BINS, 1, ft.N_FFT), torch.float32),
        ("logmel_extractor.melW", (ft.N_BINS, ft.N_MELS), torch.float32),
        ("bn0.weight", (ft.N_MELS,), torch.float32),
        ("bn0.bias", (ft.N_MELS,), torch.float32),
        ("bn0.running_mean", (ft.N_MELS,), torch.float32),
        ("bn0.running_var", (ft.N_MELS,), torch.float32),
        ("bn0.num_batches_tracked", (), torch.int64),
        ("downsample_layers.0.0.weight", (DIMS[0], 1, 4, 4), torch.float32),
        ("downsample_layers.0.0.bias", (DIMS[0],), torch.float32),
        ("downsample_layers.0.1.weight", (DIMS[0],), torch.float32),
        ("downsample_layers.0.1.bias", (DIMS[0],), torch.float32),
    ]
    for i in range(1, 4):
        spec += [
            ("downsample_layers.%d.0.weight" % i, (DIMS[i - 1],), torch.float32),
            ("downsample_layers.%d.0.bias" % i, (DIMS[i - 1],), torch.float32),
            ("downsample_layers.%d.1.weight" % i, (DIMS[i], DIMS[i - 1], 2, 2), torch.float32),
            ("downsample_layers.%d.1.bias" % i, (DIMS[i],), torch.float32),
        ]
    for s in range(4):
        C = DIMS[s]
        for j in range(DEPTHS[s]):
            p = "stages.%d.%d." % (s, j)
            spec += [
                (p + "gamma", (C,), torch.float32),
                (p + "dwconv.weight", (C, 1, 7, 7), torch.float32),
                (p + "dwconv.bias", (C,), torch.float32),
                (p + "norm.weight", (C,), torch.float32),
                (p + "norm.bias", (C,), torch.float32),
                (p + "pwconv1.weight", (4 * C, C), torch.float32),
                (p + "pwconv1.bias", (4 * C,), torch.float32),
                (p + "pwconv2.weight", (C, 4 * C), torch.float32),
                (p + "pwconv2.bias", (C,), torch.float32),
            ]
    spec += [
        ("norm.weight", (DIMS[-1],), torch.float32),
        ("norm.bias", (DIMS[-1],), torch.float32),
        ("head_audioset.weight", (NUM_CLASSES, DIMS[-1]), torch.float32),
        ("head_audioset.bias", (NUM_CLASSES,), torch.float32),
    ]
    return spec


def _fan_in(shape):
    n = 1
    for d in shape[1:]:
        n *= d
    return n


def synth_state_dict(seed=0):
    """Seeded, non-degenerate weights for all 190 keys (CPU fp32 tensors)."""
    rs = np.random.RandomState(seed)
    real, imag = ft.stft_weights()
    fixed = {
        "spectrogram_extractor.stft.conv_real.weight": real,
        "spectrogram_extractor.stft.conv_imag.weight": imag,
        "logmel_extractor.melW": ft.mel_matrix(),
    }
    sd = OrderedDict()
    for key, shape, dtype in state_dict_spec():
        if key in fixed:
            a = fixed[key]
        elif key == "bn0.num_batches_tracked":
            a = np.array(0, dtype=np.int64)
        elif key == "bn0.running_mean":
            a = rs.uniform(-60.0, -20.0, size=shape)
        elif key == "bn0.running_var":
            a = rs.uniform(50.0, 400.0, size=shape)
        elif key.endswith("gamma"):
            a = rs.uniform(0.1, 0.5, size=shape)
        elif key.endswith("bias"):
            a = 0.1 * rs.standard_normal(size=shape)
        elif len(shape) == 1:                      # LN / BN scale
            a = 1.0 + 0.1 * rs.standard_normal(size=shape)
        else:                                      # conv / linear weight
            a = rs.standard_normal(size=shape) / np.sqrt(_fan_in(shape))
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dtype).reshape(shape)
        sd[key] = t
    return sd


def stress_state_dict(seed=0):
    """Weights with the statistics of a TRAINED ConvNeXt rather than of a fresh initialisation (no checkpoint ships
    with the reference and there is no network): the cases a homogeneous Gaussian recipe cannot reach.
      * LayerNorm weights log-uniform over [1e-2, 30] (block norms and downsample norms), random sign on a tenth
      * layer scale gamma log-uniform over [1e-5, 1]
      * per block, 6 "outlier" hidden units: pwconv1 rows x 300..3000 (pre-activations in the thousands, past the
        4094 where a fixed 2^4 fp16 hidden scale would saturate) with the matching pwconv2 columns scaled down, as
        trained networks balance them; 3 input columns of pwconv1 x 100
      * depthwise-conv biases with a common offset in every other block: per-pixel |mean| / std of the tensor
        entering the LayerNorm around 20 (cancellation in any algebraic LayerNorm)
      * one downsample conv with weights spanning 6 decades across output channels
    The residual stream stays O(1..10), so the absolute tolerances of the parity tests keep their meaning."""
    sd = synth_state_dict(seed)
    rs = np.random.RandomState(seed + 4242)

    def logu(lo, hi, n):
        return np.exp(rs.uniform(np.log(lo), np.log(hi), size=n))

    def put(key, arr):
        sd[key] = torch.from_numpy(np.ascontiguousarray(arr)).to(torch.float32).reshape(sd[key].shape)

    for i in range(1, 4):
        C = DIMS[i - 1]
        put("downsample_layers.%d.0.weight" % i, logu(1e-2, 30.0, C) * np.where(rs.rand(C) < 0.1, -1.0, 1.0))
        w = sd["downsample_layers.%d.1.weight" % i].numpy().astype(np.float64)
        w /= np.sqrt(56.0)                                  # LN weights above have rms 7.5: keep the conv output O(1)
        if i == 2:
            w *= logu(1e-6, 1.0, DIMS[i])[:, None, None, None]
        put("downsample_layers.%d.1.weight" % i, w)
    for s in range(4):
        C = DIMS[s]
        for j in range(DEPTHS[s]):
            p = "stages.%d.%d." % (s, j)
            put(p + "norm.weight", logu(1e-2, 30.0, C) * np.where(rs.rand(C) < 0.1, -1.0, 1.0))
            put(p + "gamma", logu(1e-5, 1.0, C))
            w1 = sd[p + "pwconv1.weight"].numpy().astype(np.float64) / 7.5     # typical pre-activation back to O(1)
            w2 = sd[p + "pwconv2.weight"].numpy().astype(np.float64)
            rows = rs.choice(4 * C, size=6, replace=False)
            boost = logu(300.0, 3000.0, 6)
            w1[rows] *= boost[:, None]
            w2[:, rows] /= boost[None, :]
            cols = rs.choice(C, size=3, replace=False)
            w1[:, cols] *= 100.0
            put(p + "pwconv1.weight", w1)
            put(p + "pwconv2.weight", w2)
            if (s + j) % 2 == 1:
                b = sd[p + "dwconv.bias"].numpy().astype(np.float64)
                put(p + "dwconv.bias", b + 20.0 * (1.0 if j % 2 else -1.0))
    return sd


def state_dict_digest(sd):
    """sha256 over key names + raw bytes, to prove two sides hold the same weights."""
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def synth_waveforms(batch, length, seed=1234, kind="noise", device="cpu"):
    """Synthetic clips (B, L) fp32.  kind: noise (sigma 0.1 ~ -20 dBFS), sweep, silence, square."""
    if kind == "noise":
        g = torch.Generator(device="cpu").manual_seed(seed)
        x = torch.randn(batch, length, generator=g, dtype=torch.float32) * 0.1
    elif kind == "sweep":
        t = torch.arange(length, dtype=torch.float64) / ft.SAMPLE_RATE
        dur = length / ft.SAMPLE_RATE
        rows = []
        for b in range(batch):
            f0, f1 = 60.0 * (b + 1), 12000.0 / (b + 1)
            phase = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / dur * t * t)
            rows.append((0.5 * torch.sin(phase)).to(torch.float32))
        x = torch.stack(rows)
    elif kind == "silence":
        x = torch.zeros(batch, length, dtype=torch.float32)
    elif kind == "square":
        t = torch.arange(length)
        x = torch.where((t // 37) % 2 == 0, 1.0, -1.0).to(torch.float32)[None, :].repeat(batch, 1)
    else:
        raise ValueError("unknown kind %r" % kind)
    return x.contiguous().to(device)


# ---- frontend probe family (round 6, tests/golden/make_frontend_goldens.py, tests/test_gpu_frontend_edge.py) -----------------
# Signals whose STFT has bins far below the frame peak -- where the FFT kernel and the reference's dense DFT round differently
# and bn0 amplifies the difference (the thin end of the 1e-3 parity margin): tones at and between bin centres, a 10 s chirp,
# an impulse train, DC + faint noise, one full-scale click over digital silence, a loud burst over near-silence.
FRONTEND_PROBES = (
    # name, samples
    ("tone_bin_centre", 32000),      # 3125 Hz = bin 100 exactly, amplitude 0.5
    ("tone_between_bins", 32000),    # bin 100.5
    ("tone_fmin", 32000),            # 50 Hz = the mel bank's lower edge (bin 1.6)
    ("two_tones_60db", 32000),       # bin 64 at 0.5 + bin 300.25 at 0.0005
    ("impulse_train", 32000),        # 0.9 every 1000 samples
    ("dc_noise", 32000),             # DC 0.5 + N(0, 0.01^2)
    ("chirp_10s", 320000),           # 20 Hz -> 15.9 kHz linear, amplitude 0.5
    ("click_silence_10s", 320000),   # digital silence, one sample at +1.0
    ("burst_quiet_10s", 320000),     # N(0, 1e-5^2) with 10 ms of full-scale noise at 5 s
)


def frontend_probe(name):
    """(1, L) fp32 waveform of one probe; float64 arithmetic rounded once."""
    L = dict(FRONTEND_PROBES)[name]
    t = np.arange(L, dtype=np.float64)
    bin_hz = ft.SAMPLE_RATE / 1024.0

    def tone(b, amp):
        return amp * np.sin(2 * np.pi * b * bin_hz * t / ft.SAMPLE_RATE)
    if name == "tone_bin_centre":
        x = tone(100.0, 0.5)
    elif name == "tone_between_bins":
        x = tone(100.5, 0.5)
    elif name == "tone_fmin":
        x = 0.5 * np.sin(2 * np.pi * 50.0 * t / ft.SAMPLE_RATE)
    elif name == "two_tones_60db":
        x = tone(64.0, 0.5) + tone(300.25, 0.0005)
    elif name == "impulse_train":
        x = np.where(t % 1000 == 0, 0.9, 0.0)
    elif name == "dc_noise":
        x = 0.5 + 0.01 * np.random.RandomState(601).randn(L)
    elif name == "chirp_10s":
        dur = L / ft.SAMPLE_RATE
        ts = t / ft.SAMPLE_RATE
        x = 0.5 * np.sin(2 * np.pi * (20.0 * ts + 0.5 * (15900.0 - 20.0) / dur * ts * ts))
    elif name == "click_silence_10s":
        x = np.zeros(L)
        x[L // 2] = 1.0
    elif name == "burst_quiet_10s":
        rs = np.random.RandomState(602)
        x = 1e-5 * rs.randn(L)
        x[160000:160320] = np.clip(0.5 * rs.randn(320), -1.0, 1.0)
    else:
        raise ValueError("unknown probe %r" % name)
    return torch.from_numpy(x.astype(np.float32))[None, :].contiguous()
