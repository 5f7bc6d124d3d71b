"""Victims that are not ours: torch ops on a side stream while split-mode kernels run on the null stream."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
B = 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
side = torch.cuda.Stream(); null_sp = ctypes.c_void_p(0)
i = 2; Ci, Co, H, W = 192, 384, 126, 28
x = torch.randn(B, H, W, Ci, device="cuda"); out = torch.empty(B, H // 2, W // 2, Co, device="cuda"); scr = torch.empty_like(x)
v = torch.randn(8192, 4096, device="cuda")
def victims():
    a = torch.sin(v) * 2 + 1                       # elementwise, no LDS
    b = torch.softmax(v, dim=1)                    # block reductions through LDS
    c = torch.fft.rfft(v[:2048, :1024], dim=1).abs()   # rocFFT: LDS exchange
    return a, b, c
ref = victims(); torch.cuda.synchronize()
bad = [0, 0, 0]
for it in range(20):
    torch.cuda.synchronize()
    for _ in range(8): lib.acx_downsample(h, i, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scr), B, H, W, null_sp)
    with torch.cuda.stream(side):
        got = victims()
    torch.cuda.synchronize()
    for k in range(3): bad[k] += int(not torch.equal(got[k], ref[k]))
print("split downsample on the null stream; torch victims wrong (elementwise, softmax, rfft) in %s of 20 runs" % bad)
