"""Serial test: poison all LDS, then run a kernel under test; a changed result = it reads LDS it never wrote."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
poi = ctypes.CDLL("/tmp/libpoison.so"); poi.poison_launch.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
L = 320000; B = 32; T = L // 320 + 1
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
wav = synth.synth_waveforms(B, L, seed=7).cuda(); sink = torch.zeros(4, dtype=torch.int32, device="cuda"); sp = ctypes.c_void_p(0)
v = torch.randn(2048, 1024, device="cuda")
def fe():
    feat = torch.empty(B, T, 224, device="cuda"); lib.acx_logmel_bn0(h, _ffi.ptr(wav), B, L, _ffi.ptr(feat), 1, sp)
    x = torch.empty(B, 252, 56, 96, device="cuda"); lib.acx_stem_ln(h, _ffi.ptr(feat), B, T, _ffi.ptr(x), sp)
    return feat, x, torch.fft.rfft(v, dim=1).abs()
ref = fe(); torch.cuda.synchronize()
for pat in (0x00000000, 0x7fc00000, 0x7f7f7f7f, 0x77007700, 0x3f800000):
    poi.poison_launch(pat, sink.data_ptr(), sp)
    got = fe(); torch.cuda.synchronize()
    print("LDS poisoned with 0x%08x: log-mel %s, stem %s, torch rfft %s" % (pat, *["same" if torch.equal(a, b) else "DIFFERENT (max %.3g, nan %s)" % (float((a - b).abs().nan_to_num(1e9).max()), bool(torch.isnan(a).any())) for a, b in zip(got, ref)]))
