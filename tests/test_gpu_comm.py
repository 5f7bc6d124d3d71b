"""GPU: the C ABI's own collective (include/acx.h, csrc/comm.hip -- SURVEY 8b / 8e): RCCL all-gather of the per-clip outputs
for a caller that has no torch.distributed.  One GPU per test box: the communicator is a world of ONE rank here (RCCL
initialises and runs the collective; the gather of one rank is the identity) -- ranks > 1 are covered on CPU through the same
shard / gather logic with gloo (tests/test_parallel_cpu.py) and on hardware by the driver's multi-GPU bench."""
import ctypes

import pytest
import torch

from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny

pytestmark = pytest.mark.gpu


def test_c_abi_allgather_world_of_one(synth_sd):
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
    m.load_state_dict(synth_sd)
    m = m.to("cuda").eval()
    dev = torch.device("cuda", 0)
    ctx = m.native_context(dev)
    lib = _ffi.lib()
    r, w = ctypes.c_int(-1), ctypes.c_int(-1)
    _ffi.check(lib.acx_comm_info(ctx.handle, ctypes.byref(r), ctypes.byref(w)))
    assert (r.value, w.value) == (0, 1)                       # no communicator yet: a world of one
    send = torch.randn(4, 527, device="cuda")
    recv = torch.empty_like(send)
    rc = lib.acx_allgather(ctx.handle, _ffi.ptr(send), _ffi.ptr(recv), send.numel() * 4, _ffi.stream_ptr(dev))
    assert rc == -2 and b"acx_comm_init" in lib.acx_last_error()      # ACX_ERR_STATE: loud, not a silent copy
    uid = _ffi.Context.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(0, 1, uid)
    wav = synth.synth_waveforms(4, 16000, seed=21).cuda()
    out = m(wav)
    gathered = ctx.allgather(out["clipwise_logits"])          # enqueued on the current stream, behind the forward
    scene = ctx.allgather(m.forward_scene_embeddings(wav))
    torch.cuda.synchronize()
    assert gathered.shape == (4, 527) and torch.equal(gathered, out["clipwise_logits"])
    assert scene.shape == (4, 768)
    _ffi.check(lib.acx_comm_info(ctx.handle, ctypes.byref(r), ctypes.byref(w)))
    assert (r.value, w.value) == (0, 1)
    with pytest.raises(_ffi.AcxError):
        ctx.comm_init(1, 1, uid)                              # rank out of range


_RANK_SCRIPT = r"""
import ctypes, os, sys, time
import torch
rank, world, idfile, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
torch.cuda.set_device(rank)
from audioset_convnext_inf_amd import _ffi
ctx = _ffi.Context(rank)
if rank == 0:
    uid = _ffi.Context.comm_unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise SystemExit("rank %d: no unique id after 120 s" % rank)
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
ctx.comm_init(rank, world, uid)
# a rank's rows are a function of (rank, row, column) the checker can recompute without any collective of its own
rows, cols = 5, 527
send = (torch.arange(rows * cols, device="cuda", dtype=torch.float32).reshape(rows, cols) + 1000.0 * rank)
recv = ctx.allgather(send)
torch.cuda.synchronize()
torch.save(recv.cpu(), outfile)
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the C-ABI all-gather across ranks needs two GPUs (RCCL refuses two ranks on one device)")
def test_c_abi_allgather_two_ranks_fresh_processes(tmp_path):
    """rank order, the ncclUniqueId handed over BY VALUE through the hand-declared function-pointer type, and which
    librccl the dlopen binds are only visible with two ranks.  Two fresh child processes (no torch.distributed anywhere), the
    id shared through a file; every rank must hold [rank 0's rows; rank 1's rows]."""
    import os
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    idfile = str(tmp_path / "uid.bin")
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", idfile, str(tmp_path / ("out%d.pt" % r))], env=env)
             for r in range(2)]
    try:
        rcs = [p.wait(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0, 0]
    base = torch.arange(5 * 527, dtype=torch.float32).reshape(5, 527)
    want = torch.cat([base, base + 1000.0])
    for r in range(2):
        got = torch.load(str(tmp_path / ("out%d.pt" % r)))
        assert got.shape == (10, 527) and torch.equal(got, want), "rank %d holds a mis-ordered gather" % r
