#!/usr/bin/env python
"""Variable-length extraction (pytorch/extract_embeddings.py): clips of many different lengths, bucketed by
length -- mostly single-clip launches, the reference's order of work (extract_embeddings.py:64-99 runs bs = 1).  Prints clips/s
and audio-seconds per second for scene embeddings.

    python tools/extract_bench.py > profiles/rNN_extract.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth                                          # noqa: E402
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny                 # noqa: E402
from audioset_convnext_inf_amd.pytorch.extract_embeddings import extract             # noqa: E402

m = convnext_tiny(after_stem_dim=[252, 56])
m.load_state_dict(synth.synth_state_dict(0))
m = m.cuda().eval()
rs = np.random.RandomState(0)
for name, lengths in (("300 clips, every length different (1 .. 30 s)", rs.randint(32000, 960000, size=300)),
                      ("300 clips of 40 distinct lengths (1 .. 30 s)", rs.choice(rs.randint(32000, 960000, size=40), size=300)),
                      ("300 clips of 10 s", np.full(300, 320000))):
    wavs = [synth.synth_waveforms(1, int(n), seed=int(n) % 1000)[0] for n in lengths]
    extract(m, wavs[:8], what="scene")                                               # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = extract(m, wavs, what="scene")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(out) == len(wavs) and all(o.shape == (768,) for o in out)
    print("%-48s %7.1f clips/s  %8.0f s of audio per s  (%.2f s)" % (name, len(wavs) / dt, float(np.sum(lengths)) / 32000 / dt, dt))
