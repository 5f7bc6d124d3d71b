#!/usr/bin/env python
"""Where does the evaluation sweep lose time against resident batches?  (GPU box.)  Times, for batches of 256 int16 clips in pageable host
memory: the resident forward, the host staging alone, staging + H2D alone, and the sweep of pytorch/evaluate.py::forward."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch import evaluate as ev
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
from audioset_convnext_inf_amd.utils.data_generator import ClipShard, evaluate_batches

n, B, L = 4096, 256, 320000
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
g = np.random.Generator(np.random.PCG64(11))
wav = g.integers(-3277, 3277, size=(n, L), dtype=np.int16)
shard = ClipShard(wav, np.zeros((n, 527), np.bool_))
dev = torch.device("cuda", 0)
res = ev.pcm16_to_float32(torch.from_numpy(wav[:B]).cuda())
for _ in range(2): m(res)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n // B): m(res)["clipwise_output"].cpu()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("resident, scores fetched per batch: %.0f clips/s (%.2f ms per batch)" % (n / dt, 1e3 * dt / (n // B)))
st = ev.stager_for(dev)
for _ in range(5): st.to_pinned(wav[:B])
t0 = time.perf_counter()
for i in range(n // B): st.to_pinned(wav[i * B:(i + 1) * B])
dt = time.perf_counter() - t0
print("host staging alone (%d copy threads): %.2f ms per batch = %.1f GB/s" % (st.kReaders, 1e3 * dt / (n // B), wav[:B].nbytes * (n // B) / dt / 1e9))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n // B):
    d = st.to_device(st.to_pinned(wav[i * B:(i + 1) * B]))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("staging + H2D + widening, no model: %.2f ms per batch = %.1f GB/s over PCIe" % (1e3 * dt / (n // B), wav[:B].nbytes * (n // B) / dt / 1e9))
for k in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev.forward(m, evaluate_batches(shard, batch_size=B, device_cast=True))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("sweep %d: %.0f clips/s (%.2f ms per batch)" % (k, n / dt, 1e3 * dt / (n // B)))
# the same with the reader's threads pinned down to fewer: is the host copy what competes?
for thr in (2, 16):
    ev._Stager.kReaders = thr; ev._STAGERS.clear()
    ev.forward(m, evaluate_batches(shard, batch_size=B, device_cast=True))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev.forward(m, evaluate_batches(shard, batch_size=B, device_cast=True))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("sweep with %d copy threads: %.0f clips/s" % (thr, n / dt))

# ---- per-iteration timeline of the sweep loop (host clock and GPU events) ----
ev._Stager.kReaders = 8; ev._STAGERS.clear()
stage = ev.stager_for(dev)
def prepare(b): return stage.to_pinned(b["waveform"])
for rep in range(2):
    rows = []
    pending = None
    t_prev = time.perf_counter()
    gpu_ev = []
    for batch, host in ev._ahead(evaluate_batches(shard, batch_size=B, device_cast=True), prepare):
        t0 = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        x = stage.to_device(host)
        t1 = time.perf_counter()
        e0.record()
        with torch.no_grad():
            out = m(x)
        e1.record()
        t2 = time.perf_counter()
        if pending is not None:
            pending["clipwise_output"].cpu()
        t3 = time.perf_counter()
        pending = out
        gpu_ev.append((e0, e1))
        rows.append((t0 - t_prev, t1 - t0, t2 - t1, t3 - t2))
        t_prev = t3
    torch.cuda.synchronize()
    if rep == 1:
        print("per iteration, ms: wait for the staged batch | to_device | model() launches | fetch previous scores || GPU: forward, gap to the next forward's start")
        for i, r in enumerate(rows):
            fwd = gpu_ev[i][0].elapsed_time(gpu_ev[i][1])
            gap = gpu_ev[i][1].elapsed_time(gpu_ev[i + 1][0]) if i + 1 < len(rows) else 0.0
            print("  %2d  %6.2f %6.2f %6.2f %6.2f || %6.2f %6.2f" % ((i,) + tuple(1e3 * v for v in r) + (fwd, gap)))
