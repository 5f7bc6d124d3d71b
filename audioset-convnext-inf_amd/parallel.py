"""Multi-GPU data parallelism of the inference path: one process per GPU, every rank holds a full weight
replica (113 MB fp32), the batch of clips is sharded, and the ONLY collective is the all-gather of the
per-clip outputs (logits 64x527 fp32 = 135 KB per rank at bs=512/8) -- RCCL over xGMI through
torch.distributed (backend "nccl"); gloo on CPU for the tests.  The reference has no inference-time
collective at all (its only parallelism is training DDP, main.py:641,992-997); clips are independent
in eval mode (BatchNorm uses running statistics, convnext.py:219,305), so sharding is exact.

Frame embeddings (43 MB per 64 clips) stay sharded on purpose -- never gathered.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n, world_size, r):
    """Contiguous, equal-size shards (last ranks may be padded): returns (start, stop, per_rank)."""
    per = (n + world_size - 1) // world_size
    start = min(n, r * per)
    return start, min(n, start + per), per


def shard_rows(x, world_size=None, r=None, pad=True):
    """This rank's rows of a (N, ...) batch; padded with zero rows to equal shard size when `pad`
    (the latency-bound all-gather wants equal contributions)."""
    world_size = world() if world_size is None else world_size
    r = rank() if r is None else r
    start, stop, per = shard_bounds(x.shape[0], world_size, r)
    part = x[start:stop]
    if pad and part.shape[0] < per:
        fill = torch.zeros((per - part.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        part = torch.cat([part, fill], dim=0)
    return part.contiguous()


def all_gather_rows(local, total_rows=None):
    """Concatenate every rank's (n_local, ...) tensor along dim 0 in rank order (one collective, issued
    on the current stream's process group); trims padding rows when `total_rows` is given."""
    if world() == 1:
        return local if total_rows is None else local[:total_rows]
    local = local.contiguous()
    out = torch.empty((world() * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out if total_rows is None else out[:total_rows]


def sharded_forward(model, wav, what="logits"):
    """Data-parallel counterpart of `model(wav)`: every rank passes the SAME global batch (or just its own
    rows with `wav_is_local`), computes its shard and receives the gathered result."""
    n = wav.shape[0]
    part = shard_rows(wav)
    if what == "logits":
        out = model(part)
        return {k: all_gather_rows(v, n) for k, v in out.items()}
    if what == "scene":
        return all_gather_rows(model.forward_scene_embeddings(part), n)
    raise ValueError("frame embeddings stay sharded; call model.forward_frame_embeddings on the local rows")
