import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
vic = ctypes.CDLL("/tmp/libvictim.so"); vic.victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
B = 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
side = torch.cuda.Stream(); side_sp = ctypes.c_void_p(side.cuda_stream); null_sp = ctypes.c_void_p(0)
i = 2; Ci, Co, H, W = 192, 384, 126, 28
x = torch.randn(B, H, W, Ci, device="cuda"); out = torch.empty(B, H // 2, W // 2, Co, device="cuda"); scr = torch.empty_like(x)
n = 1 << 24; data = torch.randn(n, device="cuda")
def run(aggr):
    sums = torch.zeros(16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    if aggr:
        for _ in range(9): lib.acx_downsample(h, i, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scr), B, H, W, null_sp)
    vic.victim_launch(data.data_ptr(), sums.data_ptr(), 2048, 64, n, side_sp)
    torch.cuda.synchronize()
    return sums.cpu().tolist()[:9]
ref = run(False)
r2 = run(False)
print("self-check (no aggressor) equal per class:", [a == b for a, b in zip(ref, r2)])
bad = [0] * 9
for it in range(20):
    got = run(True)
    for k in range(9): bad[k] += int(got[k] != ref[k])
print("victim classes wrong (dword loads, dwordx4 loads, LDS b32, transcendental chain, shuffles, LDS b64, LDS b128, dwordx2 loads, LDS write2/read2_b32 FFT-shaped): %s of 20" % bad)
