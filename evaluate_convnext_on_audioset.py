#!/usr/bin/env python
"""AudioSet evaluation sweep on the MI355X path -- counterpart of the reference's
evaluate_convnext_on_audioset.py: bs=256 sequential batches, mAP / AUC / d-prime printed as
`Validate <set> <metric>: x.xxx`.  One process per GPU under torchrun (clips sharded by batch, scores
gathered once at the end); a single process works too.

Data: either the reference's packed HDF5 files (needs h5py) or .npy shards (int16 waveforms (N,320000) +
targets (N,527)).  Without data, --synthetic N scores a seeded synthetic set (sanity check of the plumbing
and a clips/s figure for the whole sweep including the int16 -> fp32 conversion and the H2D copy).

    torchrun --nproc-per-node 8 evaluate_convnext_on_audioset.py --tiny_path ckpt.pth --waveforms eval_wav.npy --targets eval_tgt.npy
"""
import argparse
import os
import sys
import time

# two hardware queues for this process's HIP streams (an exported value wins): the package's two sub-batch streams and the sweep's
# copy streams overlap best on two -- bench.py and INTEGRATION.md have the measurement.  Before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import numpy as np      # noqa: E402
import torch            # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny, load_checkpoint       # noqa: E402
from audioset_convnext_inf_amd.pytorch.evaluate import evaluate_sharded                     # noqa: E402
from audioset_convnext_inf_amd.utils.data_generator import ClipShard                        # noqa: E402


def synthetic_shard(n, seed=0):
    rs = np.random.RandomState(seed)
    wav = (rs.standard_normal((n, 320000)).astype(np.float32) * 0.1 * 32767.0).astype(np.int16)
    tgt = rs.uniform(size=(n, 527)) < 0.05
    tgt[:2] = [[True] * 527, [False] * 527]               # every class has a positive and a negative
    return ClipShard(wav, tgt)


def evaluate(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local_rank))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(local_rank)

    model = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                          use_speed_perturb=False)
    if rank == 0:
        print("total_params", sum(p.numel() for p in model.parameters() if p.requires_grad))
    if args.tiny_path:
        load_checkpoint(model, args.tiny_path)
    else:
        from audioset_convnext_inf_amd import synth
        model.load_state_dict(synth.synth_state_dict(0))
    model.to(torch.device("cuda", local_rank)).eval()

    sets = []
    if args.synthetic:
        sets.append(("synthetic", synthetic_shard(args.synthetic)))
    if args.waveforms:
        sets.append(("test", ClipShard.from_npy(args.waveforms, args.targets)))
    if args.hdf5:
        sets.append(("test", ClipShard.from_hdf5(args.hdf5)))
    for name, shard in sets:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = evaluate_sharded(model, shard, batch_size=args.batch_size)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            print("Validate %s mAP: %.3f" % (name, np.mean(stats["average_precision"])))
            print("Validate %s AUC: %.3f" % (name, np.mean(stats["auc"])))
            print("Validate %s d-prime: %.3f" % (name, np.mean(stats["d_prime"])))
            print("(%d clips in %.2f s on %d GPU(s): %.1f clips/s incl. int16->fp32, H2D and metrics)"
                  % (len(shard), dt, world, len(shard) / dt))


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Evaluate ConvNeXt-Tiny on AudioSet on MI355X.")
    p.add_argument("--tiny_path", type=str, default=None, help=".pth ({'model': sd}) or .safetensors checkpoint")
    p.add_argument("--waveforms", type=str, default=None)
    p.add_argument("--targets", type=str, default=None)
    p.add_argument("--hdf5", type=str, default=None)
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--batch_size", type=int, default=256)
    evaluate(p.parse_args())
