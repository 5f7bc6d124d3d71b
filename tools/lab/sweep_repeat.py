import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch import evaluate as ev
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
from audioset_convnext_inf_amd.utils.data_generator import ClipShard, evaluate_batches
n, B, L = 4096, 256, 320000
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
g = np.random.Generator(np.random.PCG64(11))
wav = g.integers(-3277, 3277, size=(n, L), dtype=np.int16)
shard = ClipShard(wav, np.zeros((n, 527), np.bool_))
half = ClipShard(wav[:2048], np.zeros((2048, 527), np.bool_))
for k in range(6):
    sh = half if k % 2 == 0 else shard
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev.forward(m, evaluate_batches(sh, batch_size=B, device_cast=True))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("sweep %d (%d clips): %.0f clips/s (%.2f ms per batch)" % (k, len(sh), len(sh) / dt, 1e3 * dt / (len(sh) // B)))
for thr in (2, 8, 16, 8, 4, 8):
    ev._Stager.kReaders = thr; ev._STAGERS.clear()
    ev.forward(m, evaluate_batches(half, batch_size=B, device_cast=True))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev.forward(m, evaluate_batches(shard, batch_size=B, device_cast=True))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("fresh stager, %2d copy threads: %.0f clips/s (%.2f ms per batch)" % (thr, n / dt, 1e3 * dt / (n // B)))
