#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE model class.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  Nothing of the reference is copied: the script imports
`audioset_convnext_inf.pytorch.convnext` from /root/reference/src after registering
stand-ins for the two third-party modules the image lacks (torchlibrosa, torchaudio -- both
are pip dependencies of the reference, not part of its source tree), installs the seeded
weights of `audioset_convnext_inf_amd.synth`, runs the reference's own
forward / forward_scene_embeddings / forward_frame_embeddings (and forward hooks for the
intermediates), and stores inputs + outputs as .npz.

The torchlibrosa stand-in is built from oracle/torchlibrosa_spec.py (restatement of the
published algorithm) with the library's module/parameter names, so the reference's call
sites (convnext.py:179-200, :298-299) run unmodified.

It also runs oracle/ref_cpu.py on the same inputs and records the deviation from the
reference in MANIFEST.json (the oracle pin).

usage: python tests/golden/make_goldens.py
"""
import json
import os
import struct
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import torchlibrosa_spec as tls          # noqa: E402
from oracle import ref_cpu                            # noqa: E402
from audioset_convnext_inf_amd import synth           # noqa: E402


# ----------------------------------------------------------------------------- stand-ins
def _install_shims():
    class _STFT(nn.Module):
        def __init__(self, n_fft, hop_length):
            super().__init__()
            out = n_fft // 2 + 1
            self.n_fft, self.hop = n_fft, hop_length
            self.conv_real = nn.Conv1d(1, out, n_fft, stride=hop_length, bias=False)
            self.conv_imag = nn.Conv1d(1, out, n_fft, stride=hop_length, bias=False)
            r, i = tls.stft_conv_weights(n_fft)
            self.conv_real.weight.data = torch.from_numpy(r)
            self.conv_imag.weight.data = torch.from_numpy(i)
            for p in self.parameters():
                p.requires_grad = False

        def forward(self, x):
            x = x[:, None, :]
            x = F.pad(x, (self.n_fft // 2, self.n_fft // 2), mode="reflect")
            real = self.conv_real(x)[:, None, :, :].transpose(2, 3)
            imag = self.conv_imag(x)[:, None, :, :].transpose(2, 3)
            return real, imag

    class Spectrogram(nn.Module):
        def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True,
                     pad_mode="reflect", power=2.0, freeze_parameters=True):
            super().__init__()
            assert (window, center, pad_mode, power) == ("hann", True, "reflect", 2.0) and win_length == n_fft
            self.stft = _STFT(n_fft, hop_length)

        def forward(self, x):
            real, imag = self.stft(x)
            return real ** 2 + imag ** 2

    class LogmelFilterBank(nn.Module):
        def __init__(self, sr=22050, n_fft=2048, n_mels=64, fmin=0.0, fmax=None, is_log=True, ref=1.0,
                     amin=1e-10, top_db=80.0, freeze_parameters=True):
            super().__init__()
            assert top_db is None
            self.ref, self.amin = ref, amin
            self.melW = nn.Parameter(torch.from_numpy(tls.mel_filterbank(sr, n_fft, n_mels, fmin, fmax).T.copy()),
                                     requires_grad=False)

        def forward(self, x):
            mel = torch.matmul(x, self.melW)
            out = 10.0 * torch.log10(torch.clamp(mel, min=self.amin, max=np.inf))
            out -= 10.0 * np.log10(np.maximum(self.amin, self.ref))
            return out

    class SpecAugmentation(nn.Module):          # training only; never called in eval
        def __init__(self, *a, **k):
            super().__init__()

    tl = types.ModuleType("torchlibrosa")
    tl_stft = types.ModuleType("torchlibrosa.stft")
    tl_aug = types.ModuleType("torchlibrosa.augmentation")
    tl_stft.Spectrogram, tl_stft.LogmelFilterBank = Spectrogram, LogmelFilterBank
    tl_aug.SpecAugmentation = SpecAugmentation
    tl.stft, tl.augmentation = tl_stft, tl_aug
    ta = types.ModuleType("torchaudio")
    ta_tr = types.ModuleType("torchaudio.transforms")
    ta_fn = types.ModuleType("torchaudio.functional")

    class Resample(nn.Module):                   # training-time augmentation only
        def __init__(self, *a, **k):
            super().__init__()
    ta_tr.Resample = Resample
    ta.transforms, ta.functional = ta_tr, ta_fn
    for name, mod in [("torchlibrosa", tl), ("torchlibrosa.stft", tl_stft), ("torchlibrosa.augmentation", tl_aug),
                      ("torchaudio", ta), ("torchaudio.transforms", ta_tr), ("torchaudio.functional", ta_fn)]:
        sys.modules[name] = mod


def _read_pcm16_wav(path):
    """Minimal RIFF reader (the sample has a LIST chunk before 'data')."""
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:4] == b"RIFF" and buf[8:12] == b"WAVE"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        body = buf[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    assert fmt[0] == 1 and fmt[1] == 1 and fmt[5] == 16, fmt
    return np.frombuffer(data, dtype="<i2").copy(), fmt[2]


def main():
    _install_shims()
    sys.path.insert(0, os.path.join(REF, "src"))
    from audioset_convnext_inf.pytorch.convnext import convnext_tiny   # the reference itself

    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                          use_speed_perturb=False)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    ref_keys = [(k, tuple(v.shape), str(v.dtype)) for k, v in model.state_dict().items()]
    sd = synth.synth_state_dict(0)
    spec_keys = [(k, tuple(s), str(d)) for k, s, d in synth.state_dict_spec()]
    assert ref_keys == spec_keys, "state_dict spec drifted from the reference"
    model.load_state_dict(sd, strict=True)
    model.eval()

    manifest = {"weights_seed": 0, "weights_sha256": synth.state_dict_digest(sd),
                "trainable_params": n_train, "state_dict_keys": len(ref_keys), "oracle_vs_reference": {},
                "shapes": {}}
    assert n_train == 28222767

    def run_ref(wav, taps=False):
        out = {}
        hooks = []
        if taps:
            def grab(name):
                def fn(mod, inp, res):
                    out[name] = res.detach().clone()
                return fn
            hooks.append(model.logmel_extractor.register_forward_hook(grab("logmel")))
            hooks.append(model.bn0.register_forward_hook(lambda m, i, r: out.__setitem__("bn0", r.transpose(1, 3).detach().clone())))
            for i in range(4):
                hooks.append(model.downsample_layers[i].register_forward_hook(grab("ds%d" % i)))
                hooks.append(model.stages[i].register_forward_hook(grab("stage%d" % i)))
                blk = model.stages[i][0]
                hooks.append(blk.dwconv.register_forward_hook(grab("s%d.b0.dwconv" % i)))
                hooks.append(blk.norm.register_forward_hook(grab("s%d.b0.ln" % i)))
                hooks.append(blk.register_forward_hook(grab("s%d.b0.out" % i)))
        with torch.no_grad():
            o = model(wav)
            out["logits"], out["probs"] = o["clipwise_logits"], o["clipwise_output"]
            for h in hooks:
                h.remove()
            out["scene"] = model.forward_scene_embeddings(wav)
            out["frame"] = model.forward_frame_embeddings(wav)
        return out

    def run_oracle(wav, taps=False):
        t = {} if taps else None
        o = ref_cpu.forward(sd, wav, taps=t)
        res = dict(t or {})
        res.update(logits=o["clipwise_logits"], probs=o["clipwise_output"],
                   scene=ref_cpu.forward_scene_embeddings(sd, wav), frame=ref_cpu.forward_frame_embeddings(sd, wav))
        return res

    def compare(tag, a, b):
        dev = {}
        for k in a:
            if k in b:
                dev[k] = float((a[k].double() - b[k].double()).abs().max())
        manifest["oracle_vs_reference"][tag] = dev
        worst = max(dev.values())
        print("  oracle vs reference [%s]: max abs dev %.3e over %d tensors" % (tag, worst, len(dev)))
        assert worst <= 1e-4, dev

    # ---- G1: the reference's demo clip (audio_samples/, PCM16 mono 32 kHz, 10 s) ----------
    pcm, sr = _read_pcm16_wav(os.path.join(REF, "audio_samples", "f62-S-v2swA_200000_210000.wav"))
    assert sr == 32000 and pcm.shape[0] == 320000
    wav = torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None, :]      # torchaudio.load convention
    r = run_ref(wav)
    compare("g1_demo", r, run_oracle(wav))
    np.savez_compressed(os.path.join(HERE, "g1_demo.npz"), pcm16=pcm,
                        **{k: v.numpy() for k, v in r.items()})
    manifest["shapes"]["g1_demo"] = {k: list(v.shape) for k, v in r.items()}

    # ---- G2: short clips with every intermediate ------------------------------------------
    L = 7680
    wav = torch.cat([synth.synth_waveforms(1, L, seed=11, kind="noise"),
                     synth.synth_waveforms(1, L, kind="sweep")])
    r = run_ref(wav, taps=True)
    compare("g2_taps", r, run_oracle(wav, taps=True))
    np.savez_compressed(os.path.join(HERE, "g2_taps.npz"), wav=wav.numpy(), **{k: v.numpy() for k, v in r.items()})
    manifest["shapes"]["g2_taps"] = {k: list(v.shape) for k, v in r.items()}

    # ---- G2b: edge-case signals, final outputs only (1 s clips: odd H at every stage) ------
    L = 32000
    wav = torch.cat([synth.synth_waveforms(1, L, seed=12, kind="noise"),
                     synth.synth_waveforms(1, L, kind="silence"),
                     synth.synth_waveforms(1, L, kind="square"),
                     synth.synth_waveforms(1, L, kind="sweep") * 1e-4])
    r = run_ref(wav)
    compare("g2_edge", r, run_oracle(wav))
    np.savez_compressed(os.path.join(HERE, "g2_edge.npz"), wav=wav.numpy(), **{k: v.numpy() for k, v in r.items()})
    manifest["shapes"]["g2_edge"] = {k: list(v.shape) for k, v in r.items()}

    # ---- G3: shape contract over clip lengths; minimum length ------------------------------
    g3 = {}
    for L in (7360, 96123, 320000, 960000):
        w = synth.synth_waveforms(1, L, seed=L)
        with torch.no_grad():
            fr = model.forward_frame_embeddings(w)
            lg = model(w)["clipwise_logits"]
        g3[str(L)] = {"frame": list(fr.shape), "logits": list(lg.shape),
                      "logits_sum": float(lg.double().sum()), "frame_abs_mean": float(fr.double().abs().mean())}
        assert list(fr.shape[2:]) == list(ref_cpu.out_hw(L)[3]), (L, fr.shape)
    try:
        with torch.no_grad():
            model(synth.synth_waveforms(1, 7359))
        g3["7359"] = "ok"
    except RuntimeError as e:
        g3["7359"] = "RuntimeError: " + str(e).split("\n")[0][:120]
    manifest["shapes"]["g3_lengths"] = g3

    # deviations measured on an MI355X per parity case (tests/parity_floor.py) live in the same file: keep them
    try:
        with open(os.path.join(HERE, "MANIFEST.json")) as f:
            kept = json.load(f).get("measured_on_mi355x")
        if kept:
            manifest["measured_on_mi355x"] = kept
    except (OSError, ValueError):
        pass
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote goldens to", HERE)


if __name__ == "__main__":
    main()
