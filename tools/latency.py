#!/usr/bin/env python
"""Small-batch latency of the hot path (eager launches vs hipGraph replay)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny

m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.cuda().eval()
for B in (1, 2, 4, 8, 16, 32, 64):
    wav = synth.synth_waveforms(B, 320000, seed=1).cuda()
    for _ in range(3): m(wav)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): m(wav)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    msg = "B=%-3d eager %.3f ms (%.0f clips/s)" % (B, dt * 1e3, B / dt)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m(wav); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = m(wav)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): g.replay()
        torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / n
        msg += " | graph %.3f ms (%.0f clips/s)" % (dg * 1e3, B / dg)
    except Exception as e:
        msg += " | graph capture failed: %s" % str(e)[:80]
    print(msg)
