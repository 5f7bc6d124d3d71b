"""Evaluation harness ("next" row 1 of SURVEY.md 8f): counterpart of reference `pytorch/evaluate.py`
(Evaluator :12-60) and `pytorch/pytorch_utils.py::forward` (:63-137): forward every batch under no_grad,
concatenate clipwise outputs and targets, then per-class average precision, ROC-AUC and
d' = sqrt(2) * Phi^-1(AUC) with exactly the sklearn / scipy calls the reference makes."""
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


def forward(model, generator, return_input=False, return_target=False):
    """Forward data to a model in mini-batches (pytorch_utils.py:63-137)."""
    output = {}
    device = next(model.parameters()).device
    model.eval()

    def append(key, value):
        output.setdefault(key, []).append(value)

    for batch in generator:
        batch_x = move_data_to_device(batch["waveform"], device)
        with torch.no_grad():
            batch_out = model(batch_x)
        append("clipwise_output", batch_out["clipwise_output"].data.cpu().numpy())
        for key in ("segmentwise_output", "framewise_output"):
            if key in batch_out:
                append(key, batch_out[key].data.cpu().numpy())
        if return_input:
            append("waveform", batch["waveform"])
        if return_target and "target" in batch:
            append("target", batch["target"])
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
    out = forward(model, evaluate_batches(shard, batch_size, rank, world), return_target=True)
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
