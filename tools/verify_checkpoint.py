#!/usr/bin/env python
"""Compare the frontend buffers a checkpoint carries with the tables this package builds (the one strong pin of
the torchlibrosa / librosa constants).

    python tools/verify_checkpoint.py <model.safetensors | checkpoint.pth> [--tol 1e-6]

Prints, for `spectrogram_extractor.stft.conv_real.weight`, `...conv_imag.weight` and `logmel_extractor.melW`, the max
|difference| against audioset-convnext-inf_amd/frontend_tables.py, says which frontend libacx would run on these buffers
(FFT when they are window x DFT within 2e-6, the dense contraction otherwise) and how many taps the banded mel filter keeps.
Exit code 0 when every deviation is <= tol, 1 otherwise, 2 when a buffer is missing.  CPU only."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import frontend_tables as ft            # noqa: E402

KEYS = ("spectrogram_extractor.stft.conv_real.weight", "spectrogram_extractor.stft.conv_imag.weight", "logmel_extractor.melW")


def load_state_dict(path):
    if str(path).endswith(".safetensors"):
        from safetensors.numpy import load_file
        return {k: np.asarray(v) for k, v in load_file(path).items()}
    import torch
    ck = torch.load(path, map_location="cpu")
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    return {k: v.detach().cpu().numpy() for k, v in sd.items() if hasattr(v, "detach")}


def report(sd, tol=1e-6, out=sys.stdout):
    missing = [k for k in KEYS if k not in sd]
    if missing:
        print("missing buffers: %s" % ", ".join(missing), file=out)
        return 2
    real, imag = ft.stft_weights()
    want = {KEYS[0]: real, KEYS[1]: imag, KEYS[2]: ft.mel_matrix()}
    worst = 0.0
    for k in KEYS:
        got = np.asarray(sd[k], dtype=np.float32)
        if got.shape != want[k].shape:
            print("%-48s shape %s, expected %s" % (k, got.shape, want[k].shape), file=out)
            return 1
        d = float(np.abs(got.astype(np.float64) - want[k].astype(np.float64)).max())
        rel = d / float(np.abs(want[k]).max())
        worst = max(worst, d)
        print("%-48s max |delta| %.3e  (%.3e of the largest entry)  %s" % (k, d, rel, "ok" if d <= tol else "DEVIATES"), file=out)
    # what libacx's acx_finalize will decide (api.hip: stft_deviation_from_dft, the window is read from bin 0)
    re, im = np.asarray(sd[KEYS[0]], np.float64)[:, 0, :], np.asarray(sd[KEYS[1]], np.float64)[:, 0, :]
    win = re[0].astype(np.float32).astype(np.float64)
    n, k = np.arange(ft.N_FFT)[None, :], np.arange(ft.N_BINS)[:, None]
    ang = 2.0 * np.pi * ((n * k) % ft.N_FFT) / ft.N_FFT
    dev = max(float(np.abs(win * np.cos(ang) - re).max()), float(np.abs(-win * np.sin(ang) - im).max()))
    mel = np.asarray(sd[KEYS[2]])
    nz = mel != 0
    taps = int(sum((np.flatnonzero(nz[:, m])[-1] - np.flatnonzero(nz[:, m])[0] + 1) if nz[:, m].any() else 0 for m in range(mel.shape[1])))
    print("STFT buffers vs window x DFT (window = bin 0): max deviation %.3e -> libacx runs the %s frontend" %
          (dev, "FFT" if dev <= 2e-6 else "dense-DFT (GEMM)"), file=out)
    print("window vs periodic hann: max |delta| %.3e" % float(np.abs(win - ft.hann()).max()), file=out)
    print("melW: %d taps in banded form (librosa's 224-bin Slaney bank: 884)" % taps, file=out)
    print("RESULT: %s (tolerance %.1e, worst %.3e)" % ("frontend tables pinned" if worst <= tol else "frontend tables DIFFER", tol, worst), file=out)
    return 0 if worst <= tol else 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint")
    ap.add_argument("--tol", type=float, default=1e-6)
    a = ap.parse_args()
    sys.exit(report(load_state_dict(a.checkpoint), a.tol))
