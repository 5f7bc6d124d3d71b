#!/usr/bin/env python
"""How much of the reference's OWN output on the frontend probes is rounding noise of its STFT formulation?

The reference evaluates the STFT as two fp32 Conv1d (convnext.py:179-187,298).  This script re-evaluates the same graph
(oracle/ref_cpu.py == the reference class bit for bit, MANIFEST.json "oracle_vs_reference") with ONE change: the two Conv1d
are accumulated in float64 and the power spectrogram is rounded to fp32 once -- the exactly rounded value of what the fp32
convolution approximates; everything downstream (mel matmul, log10, bn0, the network) is the same fp32 code.  The deviation
between the two runs is what the reference's summation order alone does to its outputs: a bin 90 dB under the frame peak holds
nothing but that noise, 10 log10 turns it into decibels and bn0 hands it on.  An implementation that sums in another order
(any FFT, any GEMM tiling, another BLAS) lands somewhere inside that cloud; agreement to better than this is luck.

Writes MANIFEST.json["reference_stft_rounding_sensitivity"][probe] = {logits, probs, scene, frame, logmel_db}; run anywhere
(CPU, no reference import).   usage: python tests/golden/frontend_self_noise.py
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_cpu                                 # noqa: E402
from audioset_convnext_inf_amd import synth               # noqa: E402


def spectrogram_f64(sd, wav):
    x = F.pad(wav[:, None, :].double(), (ref_cpu.N_FFT // 2, ref_cpu.N_FFT // 2), mode="reflect")
    real = F.conv1d(x, sd["spectrogram_extractor.stft.conv_real.weight"].double(), stride=ref_cpu.HOP)
    imag = F.conv1d(x, sd["spectrogram_extractor.stft.conv_imag.weight"].double(), stride=ref_cpu.HOP)
    return (real ** 2 + imag ** 2)[:, None].transpose(2, 3).float()


def run(sd, wav, spec_fn):
    keep = ref_cpu.spectrogram
    ref_cpu.spectrogram = spec_fn
    try:
        t = {}
        o = ref_cpu.forward(sd, wav, taps=t)
        return {"logmel_db": t["logmel"], "logits": o["clipwise_logits"], "probs": o["clipwise_output"],
                "scene": ref_cpu.forward_scene_embeddings(sd, wav), "frame": ref_cpu.forward_frame_embeddings(sd, wav)}
    finally:
        ref_cpu.spectrogram = keep


def main():
    torch.set_num_threads(8)
    sd = synth.synth_state_dict(0)
    out = {}
    for name, _ in synth.FRONTEND_PROBES:
        wav = synth.frontend_probe(name)
        a, b = run(sd, wav, ref_cpu.spectrogram), run(sd, wav, spectrogram_f64)
        out[name] = {k: float((a[k].double() - b[k].double()).abs().max()) for k in a}
        print("%-20s %s" % (name, " ".join("%s %.2e" % kv for kv in out[name].items())), flush=True)
    path = os.path.join(HERE, "MANIFEST.json")
    with open(path) as f:
        m = json.load(f)
    m["reference_stft_rounding_sensitivity"] = out
    with open(path, "w") as f:
        json.dump(m, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
