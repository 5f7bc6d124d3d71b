#!/usr/bin/env python
"""Small-batch latency of the hot path: per call p50 / p99 over many single calls (each synchronised), eager launches and
hipGraph replay, plus the host-only cost of a call (time until the launches are queued, no synchronisation).

    python tools/latency.py [clip seconds, default 10] > profiles/rNN_latency.txt

bs = 1 is the demo (demo_convnext.py) and per-file embedding extraction (extract_embeddings.py) case: there the host path
matters as much as the kernels (VERDICT r02, weak point 9)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth                                   # noqa: E402
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny          # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
L = int(32000 * secs)
m = convnext_tiny(after_stem_dim=[252, 56])
m.load_state_dict(synth.synth_state_dict(0))
m = m.cuda().eval()


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


print("# %.1f s clips @ 32 kHz, default arithmetic (%s); times in ms per call" % (secs, m.precision))
print("%-4s | %-28s | %-28s | %s" % ("B", "eager: p50 / p99 / host-only", "hipGraph replay: p50 / p99", "clips/s at p50 (eager, graph)"))
for B in (1, 2, 4, 8, 16, 64):
    wav = synth.synth_waveforms(B, L, seed=1).cuda()
    for _ in range(5):
        m(wav)
    torch.cuda.synchronize()
    n = 200 if B <= 8 else 50
    eager, host = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        m(wav)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        eager.append((time.perf_counter() - t0) * 1e3)
        host.append((t1 - t0) * 1e3)
    graph = None
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m(wav)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                m(wav)
        torch.cuda.synchronize()
        graph = []
        for _ in range(n):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            graph.append((time.perf_counter() - t0) * 1e3)
    except Exception as e:          # noqa
        print("B=%d: graph capture failed: %s" % (B, str(e)[:100]))
    e50, e99, h50 = pct(eager, 0.5), pct(eager, 0.99), pct(host, 0.5)
    if graph:
        g50, g99 = pct(graph, 0.5), pct(graph, 0.99)
        print("%-4d | %7.3f / %7.3f / %6.3f   | %7.3f / %7.3f            | %7.0f %7.0f" % (B, e50, e99, h50, g50, g99, B / e50 * 1e3, B / g50 * 1e3))
    else:
        print("%-4d | %7.3f / %7.3f / %6.3f   | -" % (B, e50, e99, h50))
