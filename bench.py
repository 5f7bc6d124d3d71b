#!/usr/bin/env python
"""Headline benchmark: clips/sec of the audio-tagging hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--mode logits|scene|frame]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, launcher form)

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself: N fresh
worker processes are spawned BEFORE this process makes any GPU call (never a re-exec), rank 0 prints the line, and
the exit code is non-zero unless all N ranks joined the RCCL group (`rccl_ranks` on the line).  `--gpus` that
disagrees with a launcher's WORLD_SIZE is an error, not a warning.

One "step" = one pass of the hot path (waveform -> logits+probs) over one batch of synthetic 10 s /
32 kHz clips that is already resident in HBM.  N = 1 runs BASELINE config 2 (ConvNeXt-Tiny, bs=64,
fp32).  The default arithmetic is "fp32_split": every fp32 GEMM operand is carried as fp16 hi + fp16 lo (representation error
<= 2^-23 relative, i.e. 23 of fp32's 24 significant bits in the worst case; the lo*lo term of a product, <= 2^-22, is dropped), each product
is three fp16 MFMAs accumulated in fp32 -- fp32-grade results (the whole GPU
parity suite runs on it at the fp32 tolerances, include/acx.h) at 16/3 of the f32-MFMA rate; the same run also
times the native f32-MFMA path (`--precision fp32`) and reports it as `native_f32_mfma`. N > 1 keeps 64 clips per GPU (weak scaling: clips are independent, every rank holds a full
weight replica) and includes the one collective of the path -- the RCCL all-gather of the logits --
in the timed region.  Rank 0 prints ONE JSON line.

The library runs a batch of 16 clips or more as two sub-batches side by side on two streams (acx_forward; `config.sub_batches`
says how many; ACX_SPLIT_STREAMS=0 turns it off): `value` times the forward as the library runs it.

Extra objects on that line:
  bf16a_shard  -- BASELINE configs[2]'s per-rank workload (64 x 10 s, set_precision("bf16a")) timed in the same process:
                  value, ms_per_step, its own kernels / roofline / roofline_dwconv (traffic from profiles/*_bf16a_traffic.json).
  frame_bs256  -- BASELINE configs[3]: forward_frame_embeddings at bs = 256: clips/s and the depthwise conv's GB/s.
                  (Both only at N = 1 with the default arithmetic; --no-extra-configs skips them.)
  eval_sweep   -- BASELINE configs[4]'s per-rank workload, PCIe included (never `value`): 4 096 int16 clips through
                  pytorch/evaluate.py::forward beside the same call on a resident batch.  Measured by a CHILD process
                  (`--eval-sweep-only`) that the default run starts and waits for before its own first GPU call.
  roofline     -- the dominant KERNEL (by device time; the event classes pw1 + pw2 are one kernel,
                  gemm_split_kernel, and are merged): algorithmic FLOPs per launch / average launch duration (HIP
                  events on the launch stream -- every launch of the profiled pass is dispatched with a start / stop event pair of
                  its own (hipExtLaunchKernel), so a duration is the dispatch's begin-to-end time, the figure rocprofv3
                  --kernel-trace reports -- taken in a separate profiled pass of the same workload so that
                  event overhead stays out of `value`; that pass runs the batch un-split on one stream, so that a launch's
                  duration is the kernel's own and not its wait for CUs the other sub-batch holds).  In fp32_split arithmetic every algorithmic fp32 flop is
                  three fp16 MFMA flops, so `peak` is the dense fp16 matrix peak / 3 (833 TFLOP/s algorithmic);
                  `frac_executed` (= frac) and `frac_algorithmic_vs_fp16_peak` (algorithmic flops against the
                  raw 2 500 TFLOP/s) are both given.
  kernels      -- the same for every kernel class, with the HBM roofline for the byte-bound ones.
  traffic      -- (inside roofline / roofline_dwconv) HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate
                  --pmc passes, tools/pmc_table.py): for the headline workload measured LIVE by this invocation -- two rocprofv3
                  child passes of tools/prof_step.py before the first GPU call of this process (`traffic_source` = "live: ...");
                  for the other arithmetics / workloads, and if the profiler is unavailable, the newest committed summary of the
                  same workload under profiles/ (`traffic_source` = its file name).
  cpu_baseline -- the CPU oracle (oracle/ref_cpu.py, torch-CPU fp32, the reference's op sequence) timed
                  on this box's host cores on a bounded sample of the same workload (N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from audioset_convnext_inf_amd import _ffi, synth                      # noqa: E402
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny   # noqa: E402

CLIP_SAMPLES = 320000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
HBM_COPY_GBS = 6290.0          # the same guide: what a float4 copy reaches on this part (79 %); SURVEY 8(d) asks for both
MFMA_F32_PEAK_TF = 157.3       # f32-input MFMA, dense (no xf32 on gfx950)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 / fp16 MFMA at the nominal 2.4 GHz
DIMS, DEPTHS = (96, 192, 384, 768), (3, 3, 9, 3)


FUSED_STAGES = (0, 1) if os.environ.get("ACX_DISABLE_FUSED_MLP", "0") != "1" else ()
# fp32_split: stage 0 runs mlp_fused_split_kernel<96> (class mlp_fused), stages 1-2 mlp_fused_wide_kernel (class mlp_wide),
# stage 3 the LayerNorm pass + two gemm_split_kernel launches per block
SPLIT_FUSED_STAGES, SPLIT_WIDE_STAGES = (0,), (1, 2)       # (the switch above acts on the native fp32 arithmetic only)
# bf16: stages 0-2 run mlp_fused_wide_bf16_kernel (class mlp_wide), stage 3 LN->bf16 rows + two gemm_bf16_kernel launches
BF16_WIDE_STAGES = (0, 1, 2)


def algorithmic_work(B, L, precision="fp32"):
    """Per kernel class: (total FLOPs, total algorithmic HBM bytes) of ONE forward.
    fp32 / fp32_split: stages 0-1 run the fused MLP kernel (hidden activation stays on chip), stages 2-3 the two
    GEMMs (split: plus the LayerNorm -> S16 pass in front of pwconv1).
    bf16: stages 0-2 run mlp_fused_wide_bf16_kernel (class mlp_wide; the hidden activation stays on chip), stage 3 is
    LN->bf16 rows (counted under rowstats), pwconv1 (bf16 in, bf16 hidden out), pwconv2."""
    T = L // 320 + 1
    hs = [(T + 4) // 4 + 1]
    ws = [56]
    for _ in range(3):
        hs.append(hs[-1] // 2)
        ws.append(ws[-1] // 2)
    pix = [B * h * w for h, w in zip(hs, ws)]
    work = {k: [0.0, 0.0] for k in _ffi.KERNEL_CLASSES}
    act = precision == "bf16a"                  # activations of stages 0-2 stored as bf16
    if act:
        precision = "bf16"
    asz = lambda s: 2 if (act and s < 3) else 4
    work["frontend"] = [0.0, B * L * 4 + B * T * 224 * 4]
    work["stem"] = [2.0 * pix[0] * 16 * 96, B * T * 224 * 4 + pix[0] * 96 * asz(0)]
    for s in range(4):
        C = DIMS[s]
        n = DEPTHS[s]
        work["dwconv"][0] += n * 2.0 * 49 * pix[s] * C
        work["dwconv"][1] += n * 2.0 * pix[s] * C * asz(s)                 # read x, write y
        if precision == "fp32_split" and (s in SPLIT_FUSED_STAGES or s in SPLIT_WIDE_STAGES):
            k = "mlp_fused" if s in SPLIT_FUSED_STAGES else "mlp_wide"
            work[k][0] += n * 4.0 * pix[s] * C * 4 * C
            work[k][1] += n * 3.0 * pix[s] * C * 4                             # y in, x in, x out
        elif precision == "fp32_split":
            work["rowstats"][1] += n * 2.0 * pix[s] * C * 4                    # LayerNorm -> S16 rows, in place
            work["pw1"][0] += n * 2.0 * pix[s] * C * 4 * C
            work["pw1"][1] += n * (pix[s] * C * 4 + pix[s] * 4 * C * 4)
            work["pw2"][0] += n * 2.0 * pix[s] * C * 4 * C
            work["pw2"][1] += n * (pix[s] * 4 * C * 4 + 2.0 * pix[s] * C * 4)
        elif precision == "bf16" and s in BF16_WIDE_STAGES:
            work["mlp_wide"][0] += n * 4.0 * pix[s] * C * 4 * C
            work["mlp_wide"][1] += n * 3.0 * pix[s] * C * asz(s)               # y in, x in, x out
        elif precision == "bf16":
            Cp = (C + 63) // 64 * 64
            work["rowstats"][1] += n * (pix[s] * C * 4 + pix[s] * Cp * 2)
            work["pw1"][0] += n * 2.0 * pix[s] * C * 4 * C
            work["pw1"][1] += n * (pix[s] * Cp * 2 + pix[s] * 4 * C * 2)
            work["pw2"][0] += n * 2.0 * pix[s] * C * 4 * C
            work["pw2"][1] += n * (pix[s] * 4 * C * 2 + 2.0 * pix[s] * C * 4)
        elif s in FUSED_STAGES:
            work["mlp_fused"][0] += n * 4.0 * pix[s] * C * 4 * C
            work["mlp_fused"][1] += n * 3.0 * pix[s] * C * 4                   # y in, x in, x out
        else:
            work["rowstats"][1] += n * 1.0 * pix[s] * C * 4
            work["pw1"][0] += n * 2.0 * pix[s] * C * 4 * C
            work["pw1"][1] += n * (pix[s] * C * 4 + pix[s] * 4 * C * 4)        # y in, hidden out
            work["pw2"][0] += n * 2.0 * pix[s] * C * 4 * C
            work["pw2"][1] += n * (pix[s] * 4 * C * 4 + 2.0 * pix[s] * C * 4)  # hidden in, x in/out
        if s > 0:
            work["downsample"][0] += 2.0 * pix[s] * 4 * DIMS[s - 1] * C
            esz = 2 if precision == "bf16" else 4
            work["downsample"][1] += pix[s - 1] * DIMS[s - 1] * esz + pix[s] * C * asz(s)
            if not ((precision == "fp32_split" and ((s - 1) in SPLIT_FUSED_STAGES or (s - 1) in SPLIT_WIDE_STAGES))
                    or (precision == "bf16" and (s - 1) in BF16_WIDE_STAGES)):
                # (in fp32_split / bf16 the last fused block of the previous stage writes the normalised rows itself)
                work["rowstats"][1] += pix[s - 1] * DIMS[s - 1] * (4 + (esz if precision != "fp32" else 0))
    work["poolhead"] = [2.0 * B * 768 * 527, pix[3] * 768 * 4]
    return work


_LIVE_TRAFFIC = {}        # (precision, workload) -> (classes, source): filled by live_traffic() before the GPU is touched


def live_traffic(precision="fp32_split"):
    """HBM bytes per launch per kernel class of the headline workload (one forward of 64 x 10 s), measured NOW: two rocprofv3 --pmc
    passes (FETCH_SIZE | WRITE_SIZE + GRBM_GUI_ACTIVE for the clock; separate passes and the gfx950 x2 correction of FETCH_SIZE as
    MI355X_MICROARCH.md prescribes, tools/pmc_table.py) of tools/prof_step.py as child processes, run to completion before this
    process makes its first GPU call (`traffic` used to be copied from profiles/ and could not contradict the
    builder's pass).  Returns (classes, source) or None when rocprofv3 is missing or a pass fails -- the committed summary of the
    same workload is then used and named as such."""
    import glob
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="acx_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for tag, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE", "GRBM_GUI_ACTIVE"])):
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", os.path.join(tmp, tag), "--", sys.executable,
                                               os.path.join(ROOT, "tools", "prof_step.py"), "--precision", precision]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0 or not glob.glob(os.path.join(tmp, tag, "*", "*counter_collection.csv")):
                print("bench.py: live PMC pass %s failed (rc %d): %s" % (tag, r.returncode, (r.stderr or "")[-300:]), file=sys.stderr)
                return None
        spec = importlib.util.spec_from_file_location("acx_pmc_table", os.path.join(ROOT, "tools", "pmc_table.py"))
        pt = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pt)
        classes = pt.traffic_summary(pt.load(os.path.join(tmp, "fetch")), pt.load(os.path.join(tmp, "write")))
        if not classes:
            return None
        return classes, "live: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE passes of tools/prof_step.py run by this bench.py invocation"
    except Exception as e:       # noqa: BLE001 -- the headline must not depend on the profiler
        print("bench.py: live PMC passes failed: %r" % (e,), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measured_traffic(precision="fp32_split", workload="bs64"):
    """HBM bytes per launch per kernel class (tools/pmc_table.py: FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes): the LIVE passes of
    this invocation when live_traffic() ran for this arithmetic and workload (the default run: fp32_split, 64 x 10 s), otherwise the
    newest committed summary of THIS arithmetic and workload ("bs64": one forward of 64 x 10 s; "frame256": forward_frame_embeddings
    at bs = 256) under profiles/.  `traffic_source` on the line says which.  The same source carries the shader clock
    (GRBM_GUI_ACTIVE / duration) the kernels of a class ran at in that pass."""
    import glob
    if (precision, workload) in _LIVE_TRAFFIC:
        return _LIVE_TRAFFIC[(precision, workload)]
    tag = {"fp32_split": "split", "bf16a": "bf16a", "bf16": "bf16", "fp32": "fp32"}[precision]
    if workload != "bs64":
        tag += "_" + workload
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_%s_traffic.json" % tag))
                   if workload != "bs64" or "_frame256_" not in os.path.basename(f))
    if not files:
        return {}, None
    try:
        with open(files[-1]) as f:
            return json.load(f)["classes"], os.path.basename(files[-1])
    except Exception:
        return {}, None


def _cpu_worker(args):
    """One process of the whole-host aggregate: `threads` torch threads, `reps` forwards of `b` clips."""
    threads, b, reps = args
    import torch as _t
    from oracle import ref_cpu
    _t.set_num_threads(threads)
    sd = synth.synth_state_dict(0)
    wav = synth.synth_waveforms(b, CLIP_SAMPLES, seed=1234)
    ref_cpu.forward(sd, wav[:1])
    t0 = time.perf_counter()
    for _ in range(reps):
        ref_cpu.forward(sd, wav)
    return b * reps, time.perf_counter() - t0


def cpu_baseline():
    """Time the oracle on the host cores (B in {1, 8, 64}, best clips/s reported).  torch-CPU scales badly
    past a few dozen threads on this graph (and the box may expose more logical CPUs than it grants), so first probe
    a few thread counts on two clips, then time B = 1 (3 reps), 8 (2 reps) and 64 (1 rep) with the best one:
    about 20-30 s of CPU work in all.  `value` is ONE process; `whole_host` adds, once, the aggregate of N such processes
    side by side (N x threads = the logical CPUs available, at most 16 processes), so that the whole-host figure is on record."""
    from oracle import ref_cpu
    sd = synth.synth_state_dict(0)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    wav = synth.synth_waveforms(64, CLIP_SAMPLES, seed=1234)
    probe = {}
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        ref_cpu.forward(sd, wav[:1])
        t0 = time.perf_counter()
        ref_cpu.forward(sd, wav[:2])
        probe[n] = 2 / (time.perf_counter() - t0)
    threads = max(probe, key=probe.get)
    torch.set_num_threads(threads)
    per_batch = {}
    for b, reps in ((1, 3), (8, 2), (64, 1)):
        best = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            ref_cpu.forward(sd, wav[:b])
            best = min(best, time.perf_counter() - t0)
        per_batch[b] = b / best
    value = max(per_batch.values())
    whole = None
    nproc = max(1, min(16, avail // threads))
    if nproc > 1:
        try:
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(nproc) as pool:
                t0 = time.perf_counter()
                res = pool.map(_cpu_worker, [(threads, 8, 1)] * nproc)
                wall = time.perf_counter() - t0
            whole = {"processes": nproc, "threads_each": threads, "clips_per_s": sum(r[0] for r in res) / max(r[1] for r in res),
                     "clips_per_s_incl_process_start": sum(r[0] for r in res) / wall, "sample": "%d processes x one forward of 8 clips each" % nproc}
        except Exception as e:                              # the baseline must never take the bench line down
            whole = {"error": repr(e)[:200]}
    return {"value": value, "unit": "clips/s", "cores": threads, "kind": "port", "processes": 1,
            "clips_per_s_by_batch": {str(k): round(v, 3) for k, v in per_batch.items()}, "whole_host": whole,
            "sample": "ONE process of oracle/ref_cpu.forward (torch-CPU fp32, the reference's op sequence) on 10 s @ 32 kHz clips at "
                      "batch 1 (best of 3), 8 (best of 2) and 64 (one pass); value = the best of the three; "
                      "thread-count probe (clips/s on 2 clips): %s; %d logical CPUs available; whole_host = N such "
                      "processes side by side, measured once" % ({k: round(v, 2) for k, v in probe.items()}, avail)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: spawn N fresh ranks (this parent has made no GPU call and makes
    none), relay rank 0's JSON line, fail unless every rank exits cleanly."""
    import subprocess
    dry = "--dry-run" in argv
    if not dry:
        have = torch.cuda.device_count()          # (the ranks are fresh child processes: nothing here is inherited by them)
        if have < n:
            print("bench.py: --gpus %d requested but only %d GPU(s) visible" % (n, have), file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Poll ALL ranks: when one dies early (bad device, import error) the others would sit in the rendezvous until the
    # process-group timeout -- stop them instead and report the first failure.  Rank 0's line is read by a thread so that
    # a full pipe never blocks it.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if any(rcs):
        print("bench.py: rank exit codes %s%s" % (rcs, "" if failed is None else " (rank %d failed first; the others were stopped)" % failed),
              file=sys.stderr)
        return 1
    return 0


def profile_rooflines(model, dev, fn, B, precision, n_prof, L=CLIP_SAMPLES):
    """A separate profiled pass of `fn` (HIP-event pairs around every launch, un-split on one stream): per-class device time,
    the roofline of the dominant matrix kernel, of the other matrix kernels, and of the depthwise conv."""
    bf16 = precision in ("bf16", "bf16a")
    split = precision == "fp32_split"
    out = {}
    ctx = model.native_context(dev)
    ctx.profile(True)
    for _ in range(n_prof):
        fn()
    torch.cuda.synchronize(dev)
    prof = ctx.profile_read()
    ctx.profile(False)
    work = algorithmic_work(B, L, precision)
    kernels = {}
    for k, (ms, n) in prof.items():
        if n == 0:
            continue
        flops, nbytes = work[k][0] * n_prof, work[k][1] * n_prof
        kernels[k] = {"launches_per_step": n // n_prof, "ms_per_step": ms / n_prof,
                      "tflops": flops / (ms * 1e-3) / 1e12 if flops else None,
                      "algorithmic_GBs": nbytes / (ms * 1e-3) / 1e9}
    out["kernels"] = kernels
    # the matrix kernels by NAME: the event classes pw1 and pw2 are launches of one kernel
    gemm_name = {"fp32": "gemm_f32_kernel", "bf16": "gemm_bf16_kernel", "bf16a": "gemm_bf16_kernel", "fp32_split": "gemm_split16_kernel"}[precision]
    fused_name = {"fp32": "mlp_fused_kernel", "bf16": "mlp_fused_bf16_kernel", "bf16a": "mlp_fused_bf16_kernel", "fp32_split": "mlp_fused_split_kernel"}[precision]
    groups = {gemm_name + " (pwconv1+GELU and pwconv2+residual launches, two-GEMM stages)": ("pw1", "pw2"),
              fused_name + " (LN+pwconv1+GELU+pwconv2+residual in one launch)": ("mlp_fused",),
              ("mlp_fused_wide_bf16_kernel (LN+pwconv1+GELU+pwconv2+residual in one launch, stages 0-2)" if bf16 else
               "mlp_fused_wide_kernel (LN+pwconv1+GELU+pwconv2+residual in one launch, stages 1-2)"): ("mlp_wide",)}
    merged = {}
    for name, ks in groups.items():
        ks = [k for k in ks if k in kernels]
        if not ks:
            continue
        n = sum(kernels[k]["launches_per_step"] for k in ks)
        merged[name] = {"ms": sum(kernels[k]["ms_per_step"] for k in ks), "launches": n,
                        "flops": sum(work[k][0] for k in ks), "bytes": sum(work[k][1] for k in ks)}
    # split mode executes 3 fp16 MFMA flops per algorithmic fp32 flop: the algorithmic peak of that arithmetic is the
    # dense fp16 peak / 3
    raw_peak = MFMA_BF16_PEAK_TF if (bf16 or split) else MFMA_F32_PEAK_TF
    mfma_mult = 3.0 if split else 1.0
    peak = raw_peak / mfma_mult
    dom = max(merged, key=lambda k: merged[k]["ms"])
    g = merged[dom]
    per_launch_flops = g["flops"] / g["launches"]
    avg_launch_s = g["ms"] * 1e-3 / g["launches"]
    ach = per_launch_flops / avg_launch_s / 1e12
    # counter traffic exists for the two workloads the PMC passes ran: 64 x 10 s (any output) and frame embeddings at bs = 256
    wl = "bs64" if (B == 64 and L == CLIP_SAMPLES) else ("frame256" if (B == 256 and L == CLIP_SAMPLES) else None)
    traffic, traffic_src = measured_traffic(precision, wl) if wl else ({}, None)
    tr = lambda k: traffic.get(k, {}).get("hbm_traffic_bytes_per_launch")
    dom_classes = [k for k in groups[dom] if k in kernels]
    tr_dom = [tr(k) for k in dom_classes]
    tr_dom = (sum(t * kernels[k]["launches_per_step"] for t, k in zip(tr_dom, dom_classes)) / g["launches"]
              if all(t is not None for t in tr_dom) else None)
    out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                       "frac": ach / peak, "frac_executed": mfma_mult * ach / raw_peak,
                       "frac_algorithmic_vs_fp16_peak" if (split or bf16) else "frac_algorithmic": ach / raw_peak,
                       "peak_note": ("dense fp16 MFMA peak %.0f TFLOP/s / %d MFMAs per fp32 product" % (raw_peak, int(mfma_mult)))
                                    if split else "dense MFMA peak of the operand type",
                       "traffic": tr_dom, "traffic_source": traffic_src, "avg_launch_ms": avg_launch_s * 1e3,
                       "launches_per_step": g["launches"], "ms_per_step": g["ms"],
                       "algorithmic_flops_per_launch": per_launch_flops,
                       "executed_mfma_flops_per_launch": mfma_mult * per_launch_flops,
                       "algorithmic_bytes_per_launch": g["bytes"] / g["launches"]}
    # the shader clock this kernel class held in the PMC pass of the same workload (GRBM_GUI_ACTIVE / 8 XCDs / duration, from
    # the traffic file -- a measurement of that run, not a constant): the matrix peak at THAT clock, beside the nominal one
    clocks = [traffic.get(k, {}).get("shader_clock_GHz") for k in dom_classes]
    if clocks and all(c is not None for c in clocks):
        ghz = sum(c * kernels[k]["ms_per_step"] for c, k in zip(clocks, dom_classes)) / sum(kernels[k]["ms_per_step"] for k in dom_classes)
        pk = raw_peak * ghz / 2.4 / mfma_mult
        out["roofline"].update({"shader_clock_GHz_in_pmc_pass": ghz, "peak_at_that_clock": pk, "frac_at_that_clock": ach / pk})
    if bf16:        # at bf16 rates a kernel may be bound by its HBM traffic, not the matrix pipe: report the binding one, keep both
        gbs = g["bytes"] / g["launches"] / avg_launch_s / 1e9
        out["roofline"].update({"mfma_tflops": ach, "mfma_frac": ach / peak, "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS})
        if gbs / HBM_PEAK_GBS > ach / peak:
            out["roofline"].update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
    out["roofline_other_matrix_kernels"] = {
        k: {"achieved": v["flops"] / (v["ms"] * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak, "ms_per_step": v["ms"], "launches_per_step": v["launches"]}
        for k, v in merged.items() if k != dom}
    mf = sum(v["flops"] for v in merged.values()) / sum(v["ms"] * 1e-3 for v in merged.values()) / 1e12
    out["roofline_all_pointwise"] = {"bound": "mfma", "achieved": mf, "peak": peak, "unit": "TFLOP/s", "frac": mf / peak}
    dw = kernels["dwconv"]
    dw_name = ("dwconv7_mfma_kernel (stages 0-2, bf16 activations) + dwconv7_col_kernel (stage 3)" if precision == "bf16a"
               else "dwconv7_col_kernel (stages 0-2) + dwconv7_tile_kernel (stage 3)")
    out["roofline_dwconv"] = {"kernel": dw_name, "bound": "hbm",
                              "achieved": dw["algorithmic_GBs"],
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dw["algorithmic_GBs"] / HBM_PEAK_GBS,
                              "frac_of_copy_rate": dw["algorithmic_GBs"] / HBM_COPY_GBS, "copy_rate": HBM_COPY_GBS,
                              "traffic": tr("dwconv"), "traffic_source": traffic_src, "ms_per_step": dw["ms_per_step"],
                              "launches_per_step": dw["launches_per_step"],
                              "algorithmic_bytes_per_launch": work["dwconv"][1] / dw["launches_per_step"]}
    return out


def timed(fn, dev, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


def eval_sweep_measure(model, dev):
    """BASELINE configs[4]'s per-rank workload, PCIe included (never `value`): 4 096 int16 clips in pageable host memory, batches of 256
    through pytorch/evaluate.py::forward as evaluate_sharded drives it (staging into the pinned ring two batches ahead, int16 over
    PCIe on a copy stream, widening on the GPU, scores fetched one batch behind) -- beside the SAME call on a batch of 256 that is
    already resident in HBM."""
    import numpy as np
    from audioset_convnext_inf_amd.pytorch import evaluate as ev
    from audioset_convnext_inf_amd.utils.data_generator import ClipShard, evaluate_batches
    n_sw = 4096
    g = np.random.Generator(np.random.PCG64(11))
    shard = ClipShard(g.integers(-3277, 3277, size=(n_sw, CLIP_SAMPLES), dtype=np.int16), np.zeros((n_sw, 527), np.bool_))
    half = ClipShard(shard.waveforms[:n_sw // 2], shard.targets[:n_sw // 2])
    res = torch.from_numpy((shard.waveforms[:256] / 32767.0).astype(np.float32)).to(dev)
    dt_res = timed(lambda: model(res)["clipwise_output"], dev, 8, 2)
    del res
    ev.forward(model, evaluate_batches(half, batch_size=256, device_cast=True))           # pins the ring once per process

    def sweep(sh):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        sc = ev.forward(model, evaluate_batches(sh, batch_size=256, device_cast=True))["clipwise_output"]
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, sc
    dt_half, _ = sweep(half)
    dt, sc = sweep(shard)
    # whole-sweep rate (pipeline fill and drain included) and the marginal rate of the second half of the clips
    # (what a rank sustains on a 20 k-clip evaluation set)
    marginal = (n_sw - n_sw // 2) / max(dt - dt_half, 1e-9)
    resident = 256 / dt_res
    return {"value": n_sw / dt, "unit": "clips/s", "clips": n_sw, "batch": 256, "seconds": dt,
            "steady_state_clips_per_s": marginal, "resident_bs256_clips_per_s": resident,
            "vs_resident_bs256": (n_sw / dt) / resident, "steady_state_vs_resident_bs256": marginal / resident,
            "scores_shape": list(sc.shape),
            "workload": "BASELINE configs[4] per-rank: int16 clips in pageable host memory -> pinned ring -> PCIe -> "
                        "acx_pcm16_to_f32 -> forward -> scores on the host (pytorch/evaluate.py::forward); PCIe-inclusive; "
                        "the resident figure beside it is the same call (waveform -> probabilities, bs = 256) on a batch already in HBM, "
                        "timed in the same process"}


def eval_sweep_child():
    """`bench.py --eval-sweep-only`: the sweep in a process of its own.  The default run starts it BEFORE the parent makes its first
    GPU call and waits for it: a process that also holds the bench's other contexts, workspaces (2.4 + 9.8 GB) and profiling events
    measured the sweep at 0.84-0.89 of the resident rate where a process that only sweeps -- what evaluate_convnext_on_audioset.py
    is -- measures 0.93-0.99 (round 5)."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
    model.load_state_dict(synth.synth_state_dict(0))
    model = model.to(dev).eval().set_precision("fp32_split")
    out = eval_sweep_measure(model, dev)
    out["measured_in"] = "a fresh child process (bench.py --eval-sweep-only), run to completion before the parent's first GPU call"
    print("ACX_EVAL_SWEEP " + json.dumps(out))


def run_eval_sweep_child():
    """-> the child's dict, or {"error": ...}; never raises (the headline must not depend on it)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--eval-sweep-only"], capture_output=True, text=True, timeout=600)
        for ln in r.stdout.splitlines():
            if ln.startswith("ACX_EVAL_SWEEP "):
                return json.loads(ln[len("ACX_EVAL_SWEEP "):])
        return {"error": "child exited %d without a result: %s" % (r.returncode, (r.stderr or "")[-400:])}
    except Exception as e:       # noqa: BLE001
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--mode", default="logits", choices=["logits", "scene", "frame"])
    ap.add_argument("--precision", default=os.environ.get("ACX_PRECISION", "fp32_split"), choices=["fp32_split", "fp32", "bf16", "bf16a"],
                    help="fp32_split (default) and fp32 both meet BASELINE configs[1]'s 1e-3 fp32 parity: split = fp32 "
                         "operands as fp16 hi+lo pairs on the fp16 matrix cores, fp32 = v_mfma_f32_32x32x2_f32; "
                         "bf16 = the arithmetic of configs[2] (bf16 contractions, fp32 LayerNorm / residual / accumulate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the bf16a_shard / frame_bs256 sub-objects")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU plumbing check of the multi-rank path (gloo, a stand-in model, no GPU): launch, rendezvous, "
                         "barriers, gather and the JSON line; the line is marked dry_run and measures nothing")
    ap.add_argument("--eval-sweep-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.eval_sweep_only:
        eval_sweep_child()
        return

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d disagrees with WORLD_SIZE %d set by the launcher" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.dry_run and os.environ.get("ACX_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        sys.exit(7)          # test hook (tests/test_host_cpu.py): a rank that dies before the rendezvous
    # HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The library drives two compute streams
    # (the sub-batches) and the evaluation sweep two more (copy in, scores out): on TWO hardware queues the sweep reaches 0.93-0.98 of
    # the resident rate where four give 0.87-0.89, and the resident forward itself is 0.7 % faster (same-box alternating runs,
    # profiles/r06_e_hw_queues.txt).  The entry scripts of the package set the same default; an exported value wins.  Must be in
    # the environment before the first HIP call of the process (and of the children below).  Multi-rank runs keep the runtime's
    # default (RCCL brings streams of its own).
    if world == 1 and not args.dry_run:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    sweep_line = None
    if (world == 1 and not args.dry_run and not args.no_profile and not args.no_extra_configs and args.mode == "logits"
            and args.batch == 64 and args.precision == "fp32_split"):
        sweep_line = run_eval_sweep_child()          # before this process touches the GPU
        live = live_traffic("fp32_split")            # likewise: counter passes as child processes
        if live is not None:
            _LIVE_TRAFFIC[("fp32_split", "bs64")] = live
    dist = None
    dev = torch.device("cpu") if args.dry_run else torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.dry_run:
            dist.init_process_group(backend="gloo", init_method="env://", world_size=world, rank=rank)
        else:
            dist.init_process_group(backend="nccl", init_method="env://", world_size=world, rank=rank, device_id=dev)
    if not args.dry_run:
        torch.cuda.set_device(dev)
    sync = (lambda: None) if args.dry_run else (lambda: torch.cuda.synchronize(dev))
    joined = 1
    if dist is not None:        # every rank must be in the group before anything is timed
        t = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        joined = int(t.item())
        if joined != world:
            print("bench.py: only %d of %d ranks joined" % (joined, world), file=sys.stderr)
            sys.exit(3)

    from audioset_convnext_inf_amd import parallel
    bf16 = args.precision in ("bf16", "bf16a")          # "bf16a": activations of stages 0-2 in HBM as bf16 too (acx.h)
    split = args.precision == "fp32_split"
    B = args.batch
    if args.dry_run:
        class _StandIn:          # per-row function of the input: exercises sharding / gather, computes nothing real
            def __call__(self, x):
                lg = x[:, :527].clone()
                return {"clipwise_output": torch.sigmoid(lg), "clipwise_logits": lg}
            def forward_scene_embeddings(self, x):
                return x[:, :768].clone()
            def forward_frame_embeddings(self, x):
                return x[:, :768 * 31 * 7].reshape(-1, 768, 31, 7).clone()
        model = _StandIn()
        wav = synth.synth_waveforms(B, CLIP_SAMPLES, seed=1234 + rank)
    else:
        model = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                              use_speed_perturb=False)
        model.load_state_dict(synth.synth_state_dict(0))
        model = model.to(dev).eval().set_precision(args.precision)
        wav = synth.synth_waveforms(B, CLIP_SAMPLES, seed=1234 + rank).to(dev)
    fn = {"logits": lambda: model(wav)["clipwise_logits"], "scene": lambda: model.forward_scene_embeddings(wav),
          "frame": lambda: model.forward_frame_embeddings(wav)}[args.mode]

    def step():
        out = fn()
        if world > 1 and args.mode != "frame":        # frame embeddings stay sharded
            out = parallel.all_gather_rows(out)
        return out

    for _ in range(args.warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if args.mode != "frame" and out.shape[0] != world * B:
            print("bench.py: gathered %d rows, expected %d" % (out.shape[0], world * B), file=sys.stderr)
            sys.exit(4)

    sub_batches = 1 if args.dry_run else model.native_context(dev).sub_batches(B)
    line = {
        "metric": "clips/sec (10 s @ 32 kHz, ConvNeXt-Tiny, bs=64)", "value": world * B * args.steps / elapsed,
        "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("bf16 (contractions and stored activations of stages 0-2)" if args.precision == "bf16a" else "bf16") if bf16 else ("f32 (GEMM operands as fp16 hi+lo pairs: representation error <= 2^-23 relative, lo*lo <= 2^-22 dropped; "
                                      "3 fp16 MFMAs per product, fp32 accumulate; all else fp32)" if split else "f32"),
        "data": "synthetic", "rccl_ranks": joined,
        "config": {"workload": "ConvNeXt-Tiny bs=%d per GPU, synthetic 10 s @ 32 kHz waveforms resident in HBM, "
                               "waveform -> %s, %s" % (B, args.mode, "bf16 contractions with fp32 LayerNorm (arithmetic of "
                                                       "BASELINE configs[2])" if bf16 else
                                                       ("fp32 via split-fp16 MFMA (BASELINE configs[1], same 1e-3 parity "
                                                        "tests as the native f32-MFMA path)" if split else
                                                        "fp32, native f32 MFMA (BASELINE configs[1])")),
                   "global_batch": world * B, "clip_samples": CLIP_SAMPLES, "weights": "seeded synthetic (synth.py)",
                   "sub_batches": sub_batches, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                   "parallelism": "clips sharded %d-way, full weight replica per GPU, RCCL all-gather of logits"
                                  % world if world > 1 else "single GPU"},
    }
    if sub_batches > 1:
        line["sub_batches_note"] = ("the library runs a batch of %d clips as %d sub-batches side by side on separate streams "
                                    "(acx_forward, ACX_SPLIT_WAYS); the per-kernel figures under `kernels` / `roofline*` come from "
                                    "a separate un-split profiled pass (launch durations that do not overlap), so their "
                                    "sum exceeds ms_per_step by what the overlap saves" % (B, sub_batches))
    if args.dry_run:
        line.update({"dry_run": True, "value": None, "ms_per_step": None,
                     "note": "plumbing check on CPU/gloo with a stand-in model: nothing was measured"})
        if rank == 0:
            print(json.dumps(line))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if rank == 0 and not args.no_profile:
        line.update(profile_rooflines(model, dev, fn, B, args.precision, max(1, min(args.steps, 5))))
    if rank == 0 and world == 1 and split and not args.no_profile:
        # the native f32-MFMA arithmetic on the same workload, same process (fewer steps: it is not the headline)
        model.set_precision("fp32")
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        n_nat = max(3, min(args.steps, 10))
        for _ in range(n_nat):
            fn()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        line["native_f32_mfma"] = {"value": B * n_nat / dt, "unit": "clips/s", "ms_per_step": 1e3 * dt / n_nat, "steps": n_nat,
                                   "note": "v_mfma_f32_32x32x2_f32 path (--precision fp32), same workload and process"}
        # the strict-fp32 arithmetic gets the same per-kernel objects (peak: the dense f32-input MFMA rate, 157.3 TFLOP/s)
        nat = profile_rooflines(model, dev, fn, B, "fp32", 2)
        line["native_f32_mfma"].update({k: nat[k] for k in ("roofline", "roofline_other_matrix_kernels", "roofline_all_pointwise", "roofline_dwconv")})
        line["native_f32_mfma"]["kernels"] = {k: {"ms_per_step": v["ms_per_step"], "launches_per_step": v["launches_per_step"]} for k, v in nat["kernels"].items()}
        model.set_precision("fp32_split")
    if rank == 0 and world == 1 and split and not args.no_profile and not args.no_extra_configs and args.mode == "logits" and B == 64:
        # The other single-GPU BASELINE configs, measured in this process next to the headline:
        #   bf16a_shard -- configs[2]'s per-rank workload (64 x 10 s, bf16 contractions + bf16 activations of stages 0-2)
        #   frame_bs256 -- configs[3]: forward_frame_embeddings at bs = 256 ("HBM-bound depthwise path": dwconv GB/s)
        model.set_precision("bf16a")
        dt = timed(fn, dev, max(20, min(args.steps, 50)), 3)
        sub = {"value": B / dt, "unit": "clips/s", "ms_per_step": 1e3 * dt, "steps": max(20, min(args.steps, 50)),
               "dtype": "bf16 (contractions and stored activations of stages 0-2; fp32 LayerNorm statistics, accumulate, GELU, residual add)",
               "workload": "BASELINE configs[2] per-rank shard: ConvNeXt-Tiny bs=64, 10 s @ 32 kHz, waveform -> logits, set_precision('bf16a')",
               "sub_batches": model.native_context(dev).sub_batches(B)}
        sub.update(profile_rooflines(model, dev, fn, B, "bf16a", 5))
        line["bf16a_shard"] = sub
        model.set_precision("fp32_split")
        B2 = 256
        wav2 = synth.synth_waveforms(B2, CLIP_SAMPLES, seed=4321).to(dev)
        fn2 = lambda: model.forward_frame_embeddings(wav2)
        dt = timed(fn2, dev, 5, 2)
        sub = {"value": B2 / dt, "unit": "clips/s", "ms_per_step": 1e3 * dt, "steps": 5,
               "workload": "BASELINE configs[3]: forward_frame_embeddings, bs=256, 10 s @ 32 kHz, output (256,768,31,7) fp32, default arithmetic",
               "sub_batches": model.native_context(dev).sub_batches(B2)}
        pr = profile_rooflines(model, dev, fn2, B2, "fp32_split", 2)
        sub["roofline_dwconv"] = pr["roofline_dwconv"]
        sub["roofline"] = pr["roofline"]
        sub["kernels"] = {k: {"ms_per_step": v["ms_per_step"], "launches_per_step": v["launches_per_step"]} for k, v in pr["kernels"].items()}
        line["frame_bs256"] = sub
        del wav2
        # configs[4]'s per-rank workload, PCIe included (never `value`): measured in a FRESH CHILD PROCESS before this process touched
        # the GPU (eval_sweep_child below), handed over here
        if sweep_line is not None:
            line["eval_sweep"] = sweep_line
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
