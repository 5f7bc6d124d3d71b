"""Needs libacx built with EXTRA=-DACX_FE_DEBUG.  Where does the log-mel kernel go wrong next to a split GEMM?"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
L = 320000; B = 32; T = L // 320 + 1; NF = B * T
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = ctypes.CDLL(_ffi.LIB_PATH); h = ctx.handle
wav = synth.synth_waveforms(B, L, seed=7).cuda()
side = torch.cuda.Stream(); side_sp = ctypes.c_void_p(side.cuda_stream); null_sp = ctypes.c_void_p(0)
i = 2; Ci, Co, H, W = 192, 384, 126, 28
x = torch.randn(B, H, W, Ci, device="cuda"); out = torch.empty(B, H // 2, W // 2, Co, device="cuda"); scr = torch.empty_like(x)
def fe(sp):
    feat = torch.empty(B, T, 224, device="cuda"); _ffi.lib().acx_logmel_bn0(h, _ffi.ptr(wav), B, L, _ffi.ptr(feat), 1, sp)
    torch.cuda.synchronize()
    dbg = np.zeros(NF * 4, dtype=np.float32); lib.acx_debug_fe_read(dbg.ctypes.data_as(ctypes.c_void_p), NF * 4)
    return feat, dbg.reshape(NF, 4).copy()
rf, rd = fe(null_sp)
for it in range(4):
    torch.cuda.synchronize()
    for _ in range(8): _ffi.lib().acx_downsample(h, i, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scr), B, H, W, null_sp)
    f, d = fe(side_sp)
    badf = ((f - rf).abs().reshape(NF, 224).amax(dim=1) > 0).cpu().numpy()
    st = [(d[:, k] != rd[:, k]) for k in range(3)]
    print("iter %d: frames with wrong log-mel %d; of those: windowed input differs %d, after butterflies differs %d, power spectrum differs %d; stage flags on OTHER frames: %s"
          % (it, badf.sum(), (st[0] & badf).sum(), (st[1] & badf).sum(), (st[2] & badf).sum(), [int((s & ~badf).sum()) for s in st]))
