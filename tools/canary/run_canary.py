import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
can = ctypes.CDLL("/tmp/libcanary.so"); can.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
what = sys.argv[1]; B = 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
side = torch.cuda.Stream(); side_sp = ctypes.c_void_p(side.cuda_stream); null_sp = ctypes.c_void_p(0)
rep = torch.zeros(4 + 4 * 64, dtype=torch.int32, device="cuda")
s = int(what[5:]) if what.startswith("block") else 2
C = (96, 192, 384, 768)[s]; H = (252, 126, 63, 31)[s]; W = (56, 28, 14, 7)[s]
need = ctypes.c_size_t(); lib.acx_block_scratch_bytes(s, B, H, W, ctypes.byref(need))
x = torch.randn(B, H, W, C, device="cuda"); scr = torch.empty(need.value, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
can.canary_launch(rep.data_ptr(), 4096, 400, side_sp)
if what != "none":
    for _ in range(10): lib.acx_block(h, s, 0, _ffi.ptr(x), B, H, W, _ffi.ptr(scr), need.value, null_sp)
torch.cuda.synchronize()
r = rep.cpu().numpy().astype("uint32")
print("%s (%s): canary mismatches %d" % (what, os.environ.get("ACX_PRECISION", "fp32_split"), r[0]))
for k in range(min(int(r[0]), 12)):
    print("   wg %d word %d (byte %d) found 0x%08x expected 0x%08x" % (r[4 + 4 * k], r[5 + 4 * k], 4 * r[5 + 4 * k], r[6 + 4 * k], r[7 + 4 * k]))
