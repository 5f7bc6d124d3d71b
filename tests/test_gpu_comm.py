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
