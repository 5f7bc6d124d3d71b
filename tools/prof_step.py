#!/usr/bin/env python
"""One warm-up + N forwards of the bench workload (for rocprofv3 --pmc / --kernel-trace passes)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth                             # noqa: E402
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--samples", type=int, default=320000)
ap.add_argument("--mode", default="logits")
ap.add_argument("--precision", default="fp32_split", choices=["fp32", "fp32_split", "bf16", "bf16a"])
a = ap.parse_args()
os.environ.setdefault("ACX_SPLIT_STREAMS", "0")      # one stream: per-dispatch counters of kernels that do not overlap
m = convnext_tiny(after_stem_dim=[252, 56])
m.load_state_dict(synth.synth_state_dict(0))
m = m.to("cuda").eval().set_precision(a.precision)
wav = synth.synth_waveforms(a.batch, a.samples, seed=1234).cuda()
for _ in range(1 + a.steps):
    if a.mode == "frame":
        m.forward_frame_embeddings(wav)
    else:
        m(wav)
torch.cuda.synchronize()
