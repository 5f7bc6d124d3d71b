#!/usr/bin/env python
"""Back-to-back steps of the bench workload (B = 64 x 10 s), eager launches against replays of one captured hipGraph:
does the GPU lose time between kernels that a graph would give back?   python tools/lab/graph_vs_eager.py [precision] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import synth                                   # noqa: E402
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny          # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32_split"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
m = convnext_tiny(after_stem_dim=[252, 56])
m.load_state_dict(synth.synth_state_dict(0))
m = m.cuda().eval().set_precision(prec)
wav = synth.synth_waveforms(64, 320000, seed=1).cuda()
for _ in range(5):
    m(wav)
torch.cuda.synchronize()

def timed(fn):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best

e = timed(lambda: m(wav))
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m(wav); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = m(wav)
torch.cuda.synchronize()
r = timed(g.replay)
e2 = timed(lambda: m(wav))
print("%s: eager %.3f ms (%.0f clips/s), graph replay %.3f ms (%.0f clips/s), eager again %.3f" % (prec, e, 64e3 / e, r, 64e3 / r, e2))
