import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
can = ctypes.CDLL("/tmp/libvcanary.so"); can.vcanary_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
B = 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
side = torch.cuda.Stream(); side_sp = ctypes.c_void_p(side.cuda_stream); null_sp = ctypes.c_void_p(0)
i = 2; Ci, Co, H, W = 192, 384, 126, 28
x = torch.randn(B, H, W, Ci, device="cuda"); out = torch.empty(B, H // 2, W // 2, Co, device="cuda"); scr = torch.empty_like(x)
for aggr in (False, True, True):
    rep = torch.zeros(4 + 4 * 64, dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
    can.vcanary_launch(rep.data_ptr(), 2048, 300, side_sp)
    if aggr:
        for _ in range(10): lib.acx_downsample(h, i, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scr), B, H, W, null_sp)
    torch.cuda.synchronize()
    r = rep.cpu().numpy().astype("uint32")
    print("aggressor %s: VGPR canary mismatches %d" % (aggr, r[0]))
    for k in range(min(int(r[0]), 8)):
        print("   wg %d reg %d lane %d found 0x%08x" % (r[4 + 4 * k], r[5 + 4 * k], r[7 + 4 * k], r[6 + 4 * k]))
