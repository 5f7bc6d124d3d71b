#!/usr/bin/env python
"""Per-step times of the bench workload (B = 64 x 10 s, waveform -> logits), one HIP event pair per step:
median, percentiles and the slowest steps -- is the mean that bench.py reports hurt by rare long steps?

    python tools/step_jitter.py [precision] [steps] > profiles/rNN_step_jitter.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth                             # noqa: E402
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny    # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32_split"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = convnext_tiny(after_stem_dim=[252, 56])
m.load_state_dict(synth.synth_state_dict(0))
m = m.cuda().eval().set_precision(prec)
wav = synth.synth_waveforms(64, 320000, seed=1234).cuda()
for _ in range(5):
    m(wav)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for a, b in ev:
    a.record()
    m(wav)
    b.record()
torch.cuda.synchronize()
t = [a.elapsed_time(b) for a, b in ev]
whole = ev[0][0].elapsed_time(ev[-1][1]) / n
s = sorted(t)
q = lambda p: s[min(n - 1, int(p * n))]
print("# %s, %d back-to-back steps (no host sync between them), ms per step" % (prec, n))
print("mean %.3f (wall / steps %.3f)  median %.3f  p10 %.3f  p90 %.3f  p99 %.3f  max %.3f" % (sum(t) / n, whole, q(0.5), q(0.1), q(0.9), q(0.99), s[-1]))
print("clips/s at the median %.0f, at the mean %.0f" % (64e3 / q(0.5), 64e3 / (sum(t) / n)))
slow = [(i, round(x, 2)) for i, x in enumerate(t) if x > 1.15 * q(0.5)]
print("steps more than 15 %% over the median: %d of %d: %s" % (len(slow), n, slow[:40]))
print("first 20 steps:", [round(x, 2) for x in t[:20]])
print("every 20th step:", [round(x, 2) for x in t[::20]])
