#!/usr/bin/env python
"""Host side of the evaluation sweep WITHOUT a GPU: how fast can R reader processes (one per rank, as under torchrun) pull
batches of a memory-mapped int16 shard into their staging buffers?  (at 8 000 clips/s a rank consumes
640 KB x 8 000 = 5.2 GB/s; eight ranks on one host need 8 x that from the page cache.)

    python tools/feed_bench.py --shard /tmp/eval_waveforms.npy --procs 8 [--batch 256] [--seconds 3]
    python tools/feed_bench.py --make 512 --shard /tmp/w.npy          # write a synthetic shard of 512 clips first

Each process runs exactly the staging step of pytorch/evaluate.py::_Stager.to_pinned (kReaders threads copying row blocks out of
the mapping into one reused buffer -- pinned on a GPU box, plain pages here) over ITS batches (rank r takes batches r, r + R,
...) round and round for --seconds (--one-pass: once, every page touched for the first time), and prints its GB/s; the parent
prints the aggregate as one JSON line.  --preadv reads the same bytes from the file into the buffer instead (measured slower:
the reason the stager has no such path).  Simultaneous rates: use a shard with at least 2 x procs batches."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(args):
    shard, rank, world, batch, seconds, preadv, one_pass = args
    from concurrent.futures import ThreadPoolExecutor
    from audioset_convnext_inf_amd.pytorch import evaluate as ev
    from audioset_convnext_inf_amd.utils.data_generator import EvaluateSampler
    w = np.load(shard, mmap_mode="r")
    readers = ev._Stager.kReaders
    pool = ThreadPoolExecutor(readers)
    dst = np.empty((batch, w.shape[1]), np.int16)
    dst[:] = 0                                     # touch the pages once, as a pinned allocation would
    fd = os.open(shard, os.O_RDONLY)
    moved, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for idx in EvaluateSampler(w.shape[0], batch, rank, world):
            x = w[int(idx[0]):int(idx[-1]) + 1]
            if not preadv:
                rows = x.shape[0]
                step = (rows + readers - 1) // readers
                list(pool.map(lambda r0: np.copyto(dst[r0:r0 + step], x[r0:r0 + step]), range(0, rows, step)))
            else:
                off = int(w.offset) + int(idx[0]) * w.shape[1] * 2
                raw = memoryview(dst[:x.shape[0]].reshape(-1).view(np.uint8))
                nbytes = x.nbytes
                stepb = -(-nbytes // readers)
                stepb += -stepb % 4096

                def read(a):
                    b = min(a + stepb, nbytes)
                    while a < b:
                        got = os.preadv(fd, [raw[a:b]], off + a)
                        if got <= 0:
                            raise IOError("short read")
                        a += got
                list(pool.map(read, range(0, nbytes, stepb)))
            moved += x.nbytes
            if time.perf_counter() - t0 >= seconds:
                break
        if one_pass:
            break
    dt = time.perf_counter() - t0
    return moved / dt / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shard", required=True)
    ap.add_argument("--make", type=int, default=0, help="write a synthetic shard of this many 10 s clips to --shard first")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--one-pass", action="store_true", help="every batch once: each page of the mapping is touched for the first time")
    ap.add_argument("--preadv", action="store_true", help="read the file into the buffer instead of copying out of the mapping (comparison)")
    a = ap.parse_args()
    if a.make:
        rs = np.random.RandomState(0)
        w = np.lib.format.open_memmap(a.shard, mode="w+", dtype=np.int16, shape=(a.make, 320000))
        for i in range(0, a.make, 64):
            w[i:i + 64] = rs.randint(-3000, 3000, size=(min(64, a.make - i), 320000)).astype(np.int16)
        w.flush()
        del w
    with mp.get_context("spawn").Pool(a.procs) as p:
        rates = p.map(worker, [(a.shard, r, a.procs, a.batch, a.seconds, a.preadv, a.one_pass) for r in range(a.procs)])
    total = float(sum(rates))
    print(json.dumps({"procs": a.procs, "batch": a.batch, "GBs_per_proc": [round(r, 2) for r in rates], "GBs_total": round(total, 2),
                      "clips_per_s_fed": round(total * 1e9 / 640000), "needed_GBs_at_8k_clips_per_rank": round(a.procs * 5.2, 1),
                      "host_cpus": os.cpu_count(), "path": "preadv into the staging buffer" if a.preadv else "copy out of the mapping (pytorch/evaluate.py)"}))


if __name__ == "__main__":
    main()
