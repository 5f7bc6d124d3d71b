"""CPU, world_size 2 over gloo: the sharding + gather plumbing of the multi-GPU path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """Stands in for the GPU model: a deterministic per-row function, so sharding must be exact."""

    def __call__(self, x):
        logits = torch.stack([x.sum(1), x.abs().sum(1), x[:, 0]], dim=1)
        return {"clipwise_output": torch.sigmoid(logits), "clipwise_logits": logits}

    def forward_scene_embeddings(self, x):
        return x[:, :4] * 2.0


def _worker(rank, world, port, n_rows, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audioset_convnext_inf_amd import parallel
    g = torch.Generator().manual_seed(0)
    wav = torch.randn(n_rows, 16, generator=g)
    full = _FakeModel()(wav)
    got = parallel.sharded_forward(_FakeModel(), wav, "logits")
    ok = all(torch.equal(got[k], full[k]) for k in full)
    sc = parallel.sharded_forward(_FakeModel(), wav, "scene")
    ok = ok and torch.equal(sc, _FakeModel().forward_scene_embeddings(wav))
    start, stop, per = parallel.shard_bounds(n_rows, world, rank)
    ok = ok and parallel.shard_rows(wav).shape[0] == per
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def _run(n_rows):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rows, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


def test_even_batch_world2():
    _run(8)


def test_ragged_batch_world2():
    _run(7)       # last shard padded, padding trimmed after the gather


def test_single_process_passthrough():
    from audioset_convnext_inf_amd import parallel
    x = torch.arange(12.0).view(6, 2)
    assert parallel.world() == 1 and parallel.rank() == 0
    assert torch.equal(parallel.all_gather_rows(x), x)
    assert parallel.shard_bounds(10, 4, 3) == (9, 10, 3)


# ---- evaluate_sharded across ranks: rank r scores batches r, r + W, ...; the last batch is short, the
#      ranks hold different numbers of batches AND of clips; one gather at the end; every rank computes the same statistics
class _FakeScorer(torch.nn.Module):
    """A deterministic per-clip scorer with the model's calling convention (dict with 'clipwise_output')."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.proj = torch.nn.Parameter(torch.randn(64, 527, generator=g), requires_grad=False)

    def forward(self, wav):
        feat = wav[:, :64 * 25].view(wav.shape[0], 64, 25).mean(2)
        logits = feat @ self.proj * 40.0
        return {"clipwise_output": torch.sigmoid(logits), "clipwise_logits": logits}


def _make_shard(n):
    import numpy as np
    from audioset_convnext_inf_amd.utils.data_generator import ClipShard
    rng = np.random.default_rng(5)
    wav = rng.integers(-20000, 20000, size=(n, 2000), dtype=np.int16)
    tgt = rng.random((n, 527)) < 0.5
    tgt[0], tgt[1] = True, False                    # every class has a positive and a negative
    return ClipShard(wav, tgt)


def _eval_worker(rank, world, port, n_clips, batch, ret):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audioset_convnext_inf_amd.pytorch import evaluate as ev
    from audioset_convnext_inf_amd.utils.data_generator import evaluate_batches
    model, shard = _FakeScorer(), _make_shard(n_clips)
    got = ev.evaluate_sharded(model, shard, batch_size=batch)
    ref = ev.Evaluator(model).evaluate(evaluate_batches(shard, batch))          # the whole sweep in this process
    ok = all(np.allclose(got[k], ref[k], rtol=0, atol=1e-12) for k in ("average_precision", "auc", "d_prime"))
    mine = sum(len(b["waveform"]) for b in evaluate_batches(shard, batch, rank, world))
    ret[rank] = (bool(ok), mine)
    dist.destroy_process_group()


def test_evaluate_sharded_world2_uneven_last_batch():
    n_clips, batch = 23, 5                      # batches of 5 5 5 5 3: rank 0 scores 13 clips (3 batches), rank 1 scores 10 (2)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, n_clips, batch, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert (ret[0][1], ret[1][1]) == (13, 10)
