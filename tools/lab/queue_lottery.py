#!/usr/bin/env python
"""Does the step time depend on WHICH hardware queue the library's sub-batch stream lands on?  N dummy streams are created before the
model's first forward (HIP hands out its hardware queues to streams round-robin), then the bs = 64 step is timed.
    python tools/lab/queue_lottery.py <N>"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.zeros(1, device="cuda")
dummies = [torch.cuda.Stream() for _ in range(n)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(1, device="cuda")        # (a stream gets its queue at first use)
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
wav = synth.synth_waveforms(64, 320000, seed=1234).cuda()
for _ in range(10): m(wav)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): m(wav)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
print("%d dummy streams first: %.3f ms per step = %.0f clips/s" % (n, 1e3 * dt, 64 / dt))
