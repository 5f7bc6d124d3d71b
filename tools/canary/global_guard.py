"""Global-memory guard: pattern-filled buffers allocated around the working buffers of split-mode kernels; any
out-of-bounds write by those kernels shows up as a changed guard word."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
B = 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
null_sp = ctypes.c_void_p(0)
def guard(): return torch.full((64 << 20,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")   # 256 MB
for s in (0, 2, 3):
    C = (96, 192, 384, 768)[s]; H = (252, 126, 63, 31)[s]; W = (56, 28, 14, 7)[s]
    need = ctypes.c_size_t(); lib.acx_block_scratch_bytes(s, B, H, W, ctypes.byref(need))
    g = [guard()]
    x = torch.randn(B, H, W, C, device="cuda"); g.append(guard())
    scr = torch.empty(need.value, dtype=torch.uint8, device="cuda"); g.append(guard())
    torch.cuda.synchronize()
    for _ in range(5): lib.acx_block(h, s, 0, _ffi.ptr(x), B, H, W, _ffi.ptr(scr), need.value, null_sp)
    torch.cuda.synchronize()
    print("block stage %d: guard words changed: %s (x at %x, scratch at %x, guards at %s)" % (
        s, [int((t != 0x5A5A5A5A).sum()) for t in g], x.data_ptr(), scr.data_ptr(), ["%x" % t.data_ptr() for t in g]))
    del g, x, scr
for i in (1, 2, 3):
    Ci = (96, 192, 384)[i - 1]; Co = (192, 384, 768)[i - 1]; H = (252, 126, 63)[i - 1]; W = (56, 28, 14)[i - 1]
    g = [guard()]; x = torch.randn(B, H, W, Ci, device="cuda"); g.append(guard())
    out = torch.empty(B, H // 2, W // 2, Co, device="cuda"); g.append(guard()); scr = torch.empty_like(x); g.append(guard())
    torch.cuda.synchronize()
    for _ in range(5): lib.acx_downsample(h, i, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scr), B, H, W, null_sp)
    torch.cuda.synchronize()
    print("downsample %d: guard words changed: %s" % (i, [int((t != 0x5A5A5A5A).sum()) for t in g]))
    del g, x, out, scr
