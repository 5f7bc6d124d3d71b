"""CPU: host-side logic -- the C-ABI library loads and exports every symbol include/acx.h declares,
geometry helpers (no GPU needed), the host emulation of the kernel's FFT schedule, the nn.Module mirror
(state_dict contract, checkpoint round trips, loud failure without a GPU)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import ConvNeXt, convnext_tiny, load_checkpoint
from oracle import ref_cpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "acx.h")).read()
    declared = set(re.findall(r"^ACX_API[^;(]*?\b(acx_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 22
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    lib = _ffi.lib()                       # binds all of them (AttributeError if one is missing)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.acx_version() >= 100
    # the header carries no torch / HIP types
    protos = "\n".join(l for l in hdr.splitlines() if l.startswith("ACX_API") or l.startswith("    "))
    assert "torch" not in protos and "hipStream_t" not in protos and "at::" not in protos


def test_geometry_helpers_match_reference_shapes():
    for L in (7360, 96123, 320000, 960000):
        assert _ffi.num_frames(L) == L // 320 + 1
        for s in range(4):
            assert _ffi.stage_hw(L, s) == tuple(ref_cpu.out_hw(L)[s])
    assert _ffi.stage_hw(320000, 0) == (252, 56)          # checkpoints/config.yaml:10-12
    assert _ffi.stage_hw(320000, 3) == (31, 7)            # README.md:61
    with pytest.raises(_ffi.AcxError, match="too short"):
        _ffi.stage_hw(7359, 3)
    n = ctypes.c_size_t()
    _ffi.check(_ffi.lib().acx_workspace_bytes(None, 64, 320000, 0, ctypes.byref(n)))
    assert 2.0e9 < n.value < 3.0e9                         # ~38 MB per clip
    assert _ffi.lib().acx_workspace_bytes(None, 0, 320000, 0, ctypes.byref(n)) == -1


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = _ffi.lib().acx_create(0, ctypes.byref(h))
    assert rc != 0 and _ffi.lib().acx_last_error()
    m = convnext_tiny(after_stem_dim=[252, 56]).eval()
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 32000))


@pytest.fixture(scope="module")
def hostfft():
    so = os.path.join(ROOT, "build", "libhostfft.so")
    if not os.path.isfile(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so,
                               os.path.join(ROOT, "tests", "host_fft_check.cpp")])
    lib = ctypes.CDLL(so)
    lib.acx_host_reflect.restype = ctypes.c_longlong
    lib.acx_host_reflect.argtypes = [ctypes.c_longlong] * 2
    return lib


def test_fft_schedule_on_host(hostfft):
    """fft_core.h (shared with the HIP kernel): 64 lanes x 3 radix-8 Stockham passes + real split."""
    rs = np.random.RandomState(0)
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1024) / 1024)).astype(np.float32)
    for amp in (0.1, 1.0, 1e-4):
        xw = ((rs.randn(1024) * amp).astype(np.float32) * w).astype(np.float32)
        P = np.zeros(513, np.float32)
        hostfft.acx_host_power_spectrum(xw.ctypes.data_as(ctypes.c_void_p), P.ctypes.data_as(ctypes.c_void_p))
        ref = np.abs(np.fft.rfft(xw.astype(np.float64))) ** 2
        assert np.abs(P - ref).max() <= 1e-6 * ref.max()
    # a pure tone lands in its bin
    xw = (np.cos(2 * np.pi * 100 * np.arange(1024) / 1024)).astype(np.float32)
    P = np.zeros(513, np.float32)
    hostfft.acx_host_power_spectrum(xw.ctypes.data_as(ctypes.c_void_p), P.ctypes.data_as(ctypes.c_void_p))
    assert P.argmax() == 100 and abs(P[100] - 512.0 ** 2) < 1.0


def test_reflect_index_matches_numpy_pad(hostfft):
    for L in (513, 1000, 7360):
        pad = np.pad(np.arange(L), (512, 512), mode="reflect")
        assert all(hostfft.acx_host_reflect(p, L) == pad[p] for p in range(L + 1024))


def test_module_contract(synth_sd):
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                      use_speed_perturb=False)
    assert isinstance(m, ConvNeXt)
    assert [(k, tuple(v.shape), v.dtype) for k, v in m.state_dict().items()] == \
           [(k, tuple(s), d) for k, s, d in synth.state_dict_spec()]
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 28222767    # README.md:49
    assert sum(p.numel() for p in m.parameters()) == 29388303
    m.load_state_dict(synth_sd, strict=True)
    with pytest.raises(ValueError):
        convnext_tiny(after_stem_dim=[56])
    with pytest.raises(NotImplementedError):
        ConvNeXt(depths=[2, 2, 8, 2], dims=[80, 160, 320, 640], num_classes=527)


def test_checkpoint_round_trips(tmp_path, synth_sd):
    from safetensors.torch import save_model
    src = convnext_tiny(after_stem_dim=[252, 56])
    src.load_state_dict(synth_sd)
    st = str(tmp_path / "model.safetensors")
    save_model(src, st)                                     # convert_pytorch_ckpt_to_safetensors.py:19
    a = ConvNeXt.from_pretrained(st, map_location="cpu")    # local-file branch, convnext.py:412-414
    pth = str(tmp_path / "convnext_tiny.pth")
    torch.save({"model": src.state_dict()}, pth)            # evaluate_convnext_on_audioset.py:36-38
    b = load_checkpoint(convnext_tiny(after_stem_dim=[252, 56]), pth)
    for k, v in synth_sd.items():
        assert torch.equal(a.state_dict()[k], v) and torch.equal(b.state_dict()[k], v)


def test_converter_script_writes_a_loadable_safetensors(tmp_path, synth_sd):
    """convert_pytorch_ckpt_to_safetensors.py (reference :1-21): .pth -> from_pretrained -> model.safetensors, which
    from_pretrained reads back to the same 190 tensors."""
    import subprocess
    import sys
    pth = str(tmp_path / "ckpt.pth")
    torch.save({"model": synth_sd}, pth)
    out = str(tmp_path / "model.safetensors")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "convert_pytorch_ckpt_to_safetensors.py"), "--ckpt", pth, "--out", out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "# params: 28222767" in r.stdout
    from audioset_convnext_inf_amd.pytorch.convnext import ConvNeXt
    m = ConvNeXt.from_pretrained(out)
    sd = m.state_dict()
    assert len(sd) == 190
    for k, v in synth_sd.items():
        assert torch.equal(sd[k], v), k


def test_bench_self_launch_dry_run():
    """`python bench.py --gpus 2` without a launcher spawns two ranks itself (before any GPU call), they rendezvous
    (gloo in the dry run), gather, and rank 0 prints the one JSON line with rccl_ranks == 2; a --gpus that disagrees
    with the launcher's WORLD_SIZE is an error."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "3", "--dry-run"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dry_run"] is True and line["value"] is None
    assert line["config"]["global_batch"] == 6
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert r.returncode == 2 and "disagrees" in r.stderr
    # a rank that dies before the rendezvous: the launcher stops the others at once instead of waiting out the
    # process-group timeout
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "2", "--dry-run"], capture_output=True, text=True, timeout=300,
                       env=dict(env, ACX_BENCH_DRYRUN_FAIL_RANK="1"))
    assert r.returncode == 1 and "rank 1 failed first" in r.stderr, r.stderr[-800:]
    assert time.time() - t0 < 120


def test_weight_signature_sees_replaced_parameters_and_submodules():
    """the per-epoch cache behind ConvNeXt._signature() must not hide a replaced Parameter or submodule
    (the native context would go on running the old packed weights), nor later in-place edits of the new tensors."""
    import torch.nn as nn
    m = convnext_tiny(after_stem_dim=[252, 56]).eval()
    s0 = m._signature()
    assert m._signature() == s0                                   # stable while nothing changes
    m.head_audioset.weight = nn.Parameter(torch.zeros(527, 768))  # parameter of a SUBmodule replaced
    s1 = m._signature()
    assert s1 != s0
    with torch.no_grad():
        m.head_audioset.weight.add_(1.0)                          # in-place edit of the NEW tensor
    s2 = m._signature()
    assert s2 != s1
    m.head_audioset = nn.Linear(768, 527)                         # whole submodule replaced
    s3 = m._signature()
    assert s3 != s2
    with torch.no_grad():
        m.head_audioset.bias.add_(1.0)                            # in-place edit inside the new submodule
    s4 = m._signature()
    assert s4 != s3 and m._signature() == s4
    m.bn0.running_mean = torch.ones(224)                          # a buffer replaced
    assert m._signature() != s4
    # the cache holds no tensors: a replaced parameter is not kept alive by it
    import weakref
    old = m.stages[0][0].gamma
    ref = weakref.ref(old)
    m.stages[0][0].gamma = nn.Parameter(torch.ones(96))
    del old
    m._signature()
    import gc
    gc.collect()
    assert ref() is None
