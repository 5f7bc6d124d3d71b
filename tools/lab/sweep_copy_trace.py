#!/usr/bin/env python
"""The sweep loop of pytorch/evaluate.py with timing events on BOTH streams: when does the copy of batch i run relative to forward i - 1?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import synth, _ffi
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
B, L, NB = 256, 320000, 10
dev = torch.device("cuda", 0)
host = [torch.empty(B * L, dtype=torch.int16, pin_memory=True) for _ in range(4)]
d16 = [torch.empty(B * L, dtype=torch.int16, device="cuda") for _ in range(4)]
d32 = [torch.empty(B, L, dtype=torch.float32, device="cuda") for _ in range(3)]
cs = torch.cuda.Stream()
comp = torch.cuda.current_stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "fetch"
for _ in range(2): m(d32[0])
torch.cuda.synchronize()
t0ev = torch.cuda.Event(enable_timing=True); t0ev.record()
rec = []
pending = None
for i in range(NB):
    s = i % 4
    c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
    f0 = torch.cuda.Event(enable_timing=True); f1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(cs):
        c0.record(cs); d16[s].copy_(host[s], non_blocking=True); c1.record(cs)
    comp.wait_event(c1)
    out32 = d32[i % 3]
    _ffi.check(_ffi.lib().acx_pcm16_to_f32(_ffi.ptr(d16[s]), _ffi.ptr(out32), B * L, _ffi.stream_ptr(dev)))
    f0.record()
    with torch.no_grad():
        out = m(out32)
    f1.record()
    if mode == "fetch" and pending is not None:
        pending["clipwise_output"].cpu()
    elif mode == "eventsync" and pending is not None:
        pending_ev.synchronize()
    pending = out; pending_ev = f1
    rec.append((c0, c1, f0, f1))
torch.cuda.synchronize()
print("mode %s: per batch [copy start, copy end | forward start, forward end] ms since t0" % mode)
for i, (c0, c1, f0, f1) in enumerate(rec):
    print("  %2d  copy %7.2f %7.2f | forward %7.2f %7.2f" % (i, t0ev.elapsed_time(c0), t0ev.elapsed_time(c1), t0ev.elapsed_time(f0), t0ev.elapsed_time(f1)))
