"""Evaluation harness ("next" row 1 of SURVEY.md 8f): counterpart of reference `pytorch/evaluate.py`
(Evaluator :12-60) and `pytorch/pytorch_utils.py::forward` (:63-137): forward every batch under no_grad,
concatenate clipwise outputs and targets, then per-class average precision, ROC-AUC and
d' = sqrt(2) * Phi^-1(AUC) with exactly the sklearn / scipy calls the reference makes."""
import threading
from math import sqrt

import numpy as np
import torch
from scipy.stats import norm
from sklearn import metrics


def move_data_to_device(x, device):
    """pytorch_utils.py:9-17, with the float32 cast the reference's waveform path forgets."""
    x = np.asarray(x)
    if x.dtype == object:
        x = np.stack([np.asarray(v, dtype=np.float32) for v in x])
    if "float" in str(x.dtype):
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    elif "int" in str(x.dtype):
        t = torch.from_numpy(np.ascontiguousarray(x)).long()
    else:
        return x
    return t.to(device, non_blocking=True)


class _Stager(object):
    """Waveform batches -> device through a ring of pinned host buffers that lives as long as the process (one stager per
    device: `stager_for`).  `to_pinned` runs in the reader thread, `to_device` in the caller: an asynchronous copy on its own
    stream, beside the model's work on the previous batch.

    How a batch reaches its pinned buffer: kReaders threads copy row blocks (numpy releases the GIL in
    copyto) -- out of the page-cache mapping for a memory-mapped .npy shard, out of memory otherwise; a batch that already IS a
    pinned torch tensor goes to the device as it is.  Measured (tools/feed_bench.py, profiles/r05_a_feed_bench.txt): the copy
    out of the mapping moves 7.5 GB/s per process on first touch and 28 GB/s on mapped pages in the build container (8 cores),
    MORE than `preadv` of the same bytes straight into the buffer (4.3 / 18 GB/s), so there is no read-into-pinned path; what
    round 4's sweep lost was not this copy but three things around it: every forward() pinned a fresh set of buffers (3 x 164 MB
    of page pinning inside a sweep of 8 batches), batches were staged one ahead only, and the int16 -> float32 widening ran as
    three element-wise torch passes over 82 M samples (3 GB of HBM traffic per batch) on the compute stream.
    int16 batches (data_generator.evaluate_batches(device_cast=True)) cross PCIe as they are and are widened on the GPU by
    the library (acx_pcm16_to_f32: float(double(x) / 32767), the arithmetic of utilities.int16_to_float32, one kernel, 0.5 GB
    of traffic per 256 clips instead of torch's three element-wise passes); float32 batches are copied as they are.  A buffer
    comes up for rewriting kRing batches later; `to_pinned` first waits for the event recorded behind the asynchronous copy
    that last READ it."""
    kReaders = 8
    kRing = 4           # two batches staged ahead + the one being staged + the one the copy engine reads

    def __init__(self, device):
        self.device = device
        self.pinned = [None] * self.kRing
        self.ready = [None] * self.kRing    # per buffer: event behind the H2D copy that read it last
        self.turn = 0
        self.copy_stream = None
        self.pool = None
        self.dev16 = [None] * self.kRing          # device-side int16 buffers, one per pinned slot
        self.dev16_read = [None] * self.kRing     # event behind the widening kernel that read it last
        self.dev32 = [None, None, None]           # float32 batches the model reads, in turn
        self.turn32 = 0
        self.fetch_stream = None                  # D2H of the scores, beside the compute stream

    def close(self):
        """Releases the pinned ring, the device buffers and the reader pool (release_stager / drop_stagers; the sweep that used them
        has been synchronised)."""
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None
        self.pinned = [None] * self.kRing
        self.ready = [None] * self.kRing
        self.dev16 = [None] * self.kRing
        self.dev16_read = [None] * self.kRing
        self.dev32 = [None, None, None]

    def staged(self, x):
        if self.device.type != "cuda":
            return False
        if isinstance(x, torch.Tensor):
            return x.dtype in (torch.int16, torch.float32) and x.dim() == 2 and x.is_pinned()
        x = np.asarray(x)
        return x.dtype in (np.int16, np.float32) and x.ndim == 2

    def _pool(self):
        if self.pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(self.kReaders)
        return self.pool

    def to_pinned(self, x):
        if isinstance(x, torch.Tensor) and x.is_pinned():
            return x, None
        n = int(np.prod(x.shape))
        slot = self.turn
        buf = self.pinned[slot]
        ev = self.ready[slot]
        if ev is not None:                  # the copy engine may still be reading this buffer
            ev.synchronize()
            self.ready[slot] = None
        want = torch.int16 if x.dtype == np.int16 else torch.float32
        if buf is None or buf.dtype != want or buf.numel() < n:
            buf = torch.empty(n, dtype=want, pin_memory=True)
            self.pinned[slot] = buf
        self.turn = (self.turn + 1) % self.kRing
        host = buf[:n].view(tuple(x.shape))
        dst = host.numpy()
        x = np.asarray(x)
        rows = x.shape[0]
        step = (rows + self.kReaders - 1) // self.kReaders
        list(self._pool().map(lambda r0: np.copyto(dst[r0:r0 + step], x[r0:r0 + step]), range(0, rows, step)))
        return host, slot

    def to_device(self, staged):
        host, slot = staged
        if self.copy_stream is None:
            # a HIGH-PRIORITY stream: HIP multiplexes streams onto a few hardware queues, and on a queue of its priority class the copy
            # of batch i + 1 sat behind the kernels of forward i (tools/sweep_timeline.py: a 3.1 ms gap between forwards = the copy)
            self.copy_stream = torch.cuda.Stream(self.device, priority=-1)
        compute = torch.cuda.current_stream(self.device)
        if host.dtype != torch.int16 or slot is None:
            with torch.cuda.stream(self.copy_stream):
                dev = host.to(self.device, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(self.copy_stream)
            if slot is not None:
                self.ready[slot] = ready
            compute.wait_event(ready)
            dev.record_stream(compute)
            return pcm16_to_float32(dev) if dev.dtype == torch.int16 else dev
        # int16 clips: device buffers of the stager's own (one int16 buffer per pinned slot, two float32 buffers the model reads in
        # turn), allocated once.  Going through the caching allocator instead -- a 164 MB block made on the copy stream and freed on
        # the compute stream every batch -- cost the sweep 4 ms per batch of 256: the H2D copy did not overlap the model at all
        # (tools/sweep_timeline.py: 35.4 ms per batch against 31.2 resident, staging 1.9 ms and copy 3.1 ms of their own).
        n = host.numel()
        d16 = self.dev16[slot]
        if d16 is None or d16.numel() < n:
            # (allocated on the CALLER's stream like d32 below, used on the copy stream behind explicit events: a replaced buffer
            # goes back to the caching allocator only after the sweep's final synchronise has drained both streams -- forward())
            if d16 is not None:
                d16.record_stream(self.copy_stream)
            d16 = self.dev16[slot] = torch.empty(n, dtype=torch.int16, device=self.device)
        t = self.turn32
        self.turn32 = (t + 1) % len(self.dev32)
        d32 = self.dev32[t]
        if d32 is None or d32.numel() < n:
            if d32 is not None:
                d32.record_stream(compute)
            d32 = self.dev32[t] = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.stream(self.copy_stream):
            if self.dev16_read[slot] is not None:           # the widening kernel that last read this buffer
                self.copy_stream.wait_event(self.dev16_read[slot])
            d16[:n].copy_(host.view(-1), non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        self.ready[slot] = ready
        compute.wait_event(ready)
        out = d32[:n].view(host.shape)
        from .. import _ffi
        _ffi.check(_ffi.lib().acx_pcm16_to_f32(_ffi.ptr(d16), _ffi.ptr(out), n, _ffi.stream_ptr(self.device)))
        done = torch.cuda.Event()
        done.record(compute)
        self.dev16_read[slot] = done
        return out          # (a view of a buffer that comes up again two batches later, in stream order behind this batch's forward)

    def plain(self, x):
        x = np.asarray(x)
        if x.dtype == np.int16:
            return torch.from_numpy((x / 32767.0).astype(np.float32)).to(self.device)
        return move_data_to_device(x, self.device)


def pcm16_to_float32(dev_int16):
    """int16 device tensor -> float32 by the library's kernel (include/acx.h: acx_pcm16_to_f32), on the current stream."""
    from .. import _ffi
    out = torch.empty(dev_int16.shape, dtype=torch.float32, device=dev_int16.device)
    _ffi.check(_ffi.lib().acx_pcm16_to_f32(_ffi.ptr(dev_int16), _ffi.ptr(out), dev_int16.numel(), _ffi.stream_ptr(dev_int16.device)))
    return out


_STAGERS = {}                    # device key -> _Stager; guarded by _STAGERS_LOCK
_STAGERS_LOCK = threading.Lock()
kAheadDepth = 2                  # batches the reader thread runs ahead of the consumer (_ahead)


def _stager_key(device):
    return (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0))


def stager_for(device):
    """The stager of `device` for ONE sweep at a time: its pinned ring is allocated once per process (pinning 4 x 164 MB costs tens
    of milliseconds -- a sweep of a few batches would spend a third of its time there if every forward() made its own).  A stager
    is checked OUT of the per-device cache for the duration of a sweep and handed back by release_stager(); a second sweep on the
    same device at the same time (another thread) gets a fresh one, which replaces the cached one when it comes back -- so the
    cache holds at most one stager per device whatever the number or identity of threads (the round-5 cache was keyed
    by thread id and never released)."""
    with _STAGERS_LOCK:
        st = _STAGERS.pop(_stager_key(device), None)
    return st if st is not None else _Stager(device)


def release_stager(stage):
    """Hands a stager back (end of a sweep).  The sweep has drained: nothing reads its buffers any more."""
    with _STAGERS_LOCK:
        old = _STAGERS.get(_stager_key(stage.device))
        _STAGERS[_stager_key(stage.device)] = stage
    if old is not None and old is not stage:
        old.close()


def drop_stagers():
    """Frees every cached stager (pinned ring, device buffers, reader pool): for long-lived processes that are done sweeping."""
    with _STAGERS_LOCK:
        olds = list(_STAGERS.values())
        _STAGERS.clear()
    for st in olds:
        st.close()


def _ahead(generator, prepare, depth=kAheadDepth):
    """Runs `generator` (and `prepare` on each item) in a background thread, `depth` batches ahead: reading the clips
    (memory-mapped shards), any host-side conversion and the copy into pinned memory overlap the GPU's work on the previous
    batch.  Exceptions re-raise in the consumer."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    done = object()
    stop = threading.Event()            # set when the consumer leaves early (exception, break, generator closed)

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def work():
        try:
            for item in generator:
                if not put((item, prepare(item))):
                    return
            put(done)
        except BaseException as e:      # noqa: B902 -- handed to the consumer
            put(e)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is done:
                break
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        t.join(timeout=10)


def forward(model, generator, return_input=False, return_target=False):
    """Forward data to a model in mini-batches (pytorch_utils.py:63-137).  Same results in the same order; the batches are
    read and staged into pinned memory two ahead in a background thread, copied to the device on a side stream, and the
    outputs of batch i are fetched only after batch i + 1 has been handed to the GPU: host work, PCIe and the model overlap."""
    output = {}
    device = next(model.parameters()).device
    model.eval()
    stage = stager_for(device)

    def append(key, value):
        output.setdefault(key, []).append(value)

    def fetch(t, done):
        """Device tensor -> numpy WITHOUT going through the compute stream: a `.cpu()` there queues behind the forward of the NEXT
        batch, which is already on that stream -- the host then runs one batch late and every H2D copy is issued only when the
        forward it should have overlapped has ended (tools/lab/sweep_copy_trace.py: 3.1 ms per batch of 256).  The copy runs on a
        side stream behind the event recorded right after the batch's own forward."""
        if device.type != "cuda":
            return t.data.cpu().numpy()
        if stage.fetch_stream is None:
            stage.fetch_stream = torch.cuda.Stream(device, priority=-1)    # (a queue of its own class: see _Stager.to_device)
        fs = stage.fetch_stream
        fs.wait_event(done)
        with torch.cuda.stream(fs):
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t.data, non_blocking=True)
            t.record_stream(fs)
            e = torch.cuda.Event()
            e.record(fs)
        e.synchronize()
        return h.numpy().copy()

    def collect(pending):
        batch, batch_out, done = pending
        append("clipwise_output", fetch(batch_out["clipwise_output"], done))
        for key in ("segmentwise_output", "framewise_output"):
            if key in batch_out:
                append(key, fetch(batch_out[key], done))
        if return_input:
            w = batch["waveform"]
            append("waveform", w if np.asarray(w).dtype != np.int16 else (np.asarray(w) / 32767.0).astype(np.float32))
        if return_target and "target" in batch:
            append("target", batch["target"])

    def prepare(batch):                     # reader thread
        return stage.to_pinned(batch["waveform"]) if stage.staged(batch["waveform"]) else None

    def device_input(batch, host):
        return stage.to_device(host) if host is not None else stage.plain(batch["waveform"])

    # The H2D copy of batch i + 1 is issued as soon as forward i has been queued -- a whole forward before its data is needed -- so
    # that a copy that lands on a hardware queue behind the library's own sub-batch stream still arrives in time.
    pending = None
    # a pinned slot comes up again kRing batches later: the reader may be `depth` ahead, one batch is being staged and one is being
    # read by the copy engine
    assert _Stager.kRing >= kAheadDepth + 2, "pinned ring shorter than the read-ahead depth + 2"
    it = _ahead(generator, prepare)
    cur = next(it, None)
    x_cur = device_input(*cur) if cur is not None else None
    while cur is not None:
        batch = cur[0]
        with torch.no_grad():
            batch_out = model(x_cur)
        done = None
        if device.type == "cuda":
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(device))
        cur = next(it, None)
        x_cur = device_input(*cur) if cur is not None else None
        if pending is not None:
            collect(pending)                # waits for the PREVIOUS batch only (its own event, a side stream)
        pending = (batch, batch_out, done)
    try:
        if pending is not None:
            collect(pending)
    finally:
        if device.type == "cuda":
            torch.cuda.synchronize(device)      # the sweep has drained: the stager's buffers are idle when the next sweep takes them
        release_stager(stage)
    return {k: np.concatenate(v, axis=0) for k, v in output.items()}


def calculate_statistics(target, clipwise_output):
    """evaluate.py:44-58."""
    average_precision = metrics.average_precision_score(target, clipwise_output, average=None)
    auc = metrics.roc_auc_score(target, clipwise_output, average=None)
    return {"average_precision": average_precision, "auc": auc, "d_prime": sqrt(2) * norm.ppf(auc)}


class Evaluator(object):
    def __init__(self, model, use_torchaudio=False):
        if use_torchaudio:
            raise NotImplementedError("Kaldi-fbank input is outside the inference contract")
        self.model = model

    def evaluate(self, data_loader):
        """-> {'average_precision': (classes,), 'auc': (classes,), 'd_prime': (classes,)}"""
        out = forward(self.model, data_loader, return_target=True)
        return calculate_statistics(out["target"], out["clipwise_output"])


def evaluate_sharded(model, shard, batch_size=256):
    """Multi-GPU sweep: rank r scores batches r, r+W, ...; scores and targets are gathered ONCE
    at the end (all_gather of padded per-rank blocks), every rank then computes the same statistics."""
    import torch.distributed as dist
    from ..utils.data_generator import evaluate_batches
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    out = forward(model, evaluate_batches(shard, batch_size, rank, world, device_cast=True), return_target=True)
    scores, target = out.get("clipwise_output"), out.get("target")
    if world > 1:
        device = next(model.parameters()).device
        nb = (len(shard) + batch_size - 1) // batch_size
        per_rank = ((nb + world - 1) // world) * batch_size
        def pad(a):
            buf = torch.zeros((per_rank, a.shape[1]), dtype=torch.float32, device=device)
            buf[: a.shape[0]] = torch.from_numpy(a).to(device)
            return buf
        n_local = torch.tensor([0 if scores is None else scores.shape[0]], device=device)
        counts = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(counts, n_local)
        classes = 527
        s_all = torch.empty((world * per_rank, classes), device=device)
        t_all = torch.empty((world * per_rank, classes), device=device)
        dist.all_gather_into_tensor(s_all, pad(scores if scores is not None else np.zeros((0, classes), np.float32)))
        dist.all_gather_into_tensor(t_all, pad(target if target is not None else np.zeros((0, classes), np.float32)))
        s_all, t_all = s_all.view(world, per_rank, classes).cpu().numpy(), t_all.view(world, per_rank, classes).cpu().numpy()
        scores = np.concatenate([s_all[r, : int(counts[r])] for r in range(world)])
        target = np.concatenate([t_all[r, : int(counts[r])] for r in range(world)])
    return calculate_statistics(target, scores)
