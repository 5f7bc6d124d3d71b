#!/usr/bin/env python
"""Does a pinned-host -> device copy on a side stream run WHILE the forward runs on the current stream?  (GPU box.)
Prints, relative to the start of a forward of 256 clips: when the copy of 164 MB issued at that moment starts and ends."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
B, L = 256, 320000
x = torch.randn(B, L, device="cuda") * 0.1
host = torch.empty(B * L, dtype=torch.int16, pin_memory=True)
dst = torch.empty(B * L, dtype=torch.int16, device="cuda")
for _ in range(2): m(x)
torch.cuda.synchronize()
def run(prio, chunks, label):
    cs = torch.cuda.Stream(priority=prio)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    ev[0].record()                      # forward starts
    m(x)
    ev[1].record()                      # forward ends
    with torch.cuda.stream(cs):
        ev[2].record(cs)
        n = B * L // chunks
        for c in range(chunks):
            dst[c * n:(c + 1) * n].copy_(host[c * n:(c + 1) * n], non_blocking=True)
        ev[3].record(cs)
    torch.cuda.synchronize()
    print("%-42s forward %.2f ms; copy from %.2f to %.2f ms after the forward's start" % (label, ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[2]), ev[0].elapsed_time(ev[3])))
run(0, 1, "side stream, one copy")
run(-1, 1, "high-priority side stream, one copy")
run(0, 16, "side stream, 16 chunks")
# the copy issued BEFORE the forward is queued
cs = torch.cuda.Stream()
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
torch.cuda.synchronize()
with torch.cuda.stream(cs):
    e[2].record(cs); dst.copy_(host, non_blocking=True); e[3].record(cs)
e[0].record(); m(x); e[1].record()
torch.cuda.synchronize()
print("%-42s forward %.2f ms; copy from %.2f to %.2f ms relative to the forward's start" % ("copy issued first", e[0].elapsed_time(e[1]), e[0].elapsed_time(e[2]), e[0].elapsed_time(e[3])))
