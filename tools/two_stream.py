#!/usr/bin/env python
"""Experiment: one batch of 64 on one stream vs sub-batches on concurrent streams (do an MFMA-bound kernel of
one part and an HBM-bound kernel of another co-run?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny

sd = synth.synth_state_dict(0)
def mk():
    m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(sd); return m.cuda().eval()
models = [mk() for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
wav = synth.synth_waveforms(128, 320000, seed=1).cuda()
def run(parts, total, n):
    per = total // parts
    chunks = [wav[i * per:(i + 1) * per].contiguous() for i in range(parts)]
    def fn(k):
        for _ in range(k):
            for i in range(parts):
                with torch.cuda.stream(streams[i]): models[i](chunks[i])
    fn(3); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%d stream(s) x %-3d  %.2f ms per %d clips (%.0f clips/s)" % (parts, per, dt * 1e3, total, total / dt))
for parts, total in ((1, 64), (2, 64), (4, 64), (1, 128), (2, 128), (4, 128), (1, 64), (2, 64)):
    run(parts, total, 15)
