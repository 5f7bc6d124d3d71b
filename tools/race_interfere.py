"""Does a kernel class running on the null stream corrupt frontend+stem results computed concurrently on a side stream?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
what = sys.argv[1]          # block0 | block2 | down2 | none
L = 320000; B = 32; T = L // 320 + 1
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
wav = synth.synth_waveforms(B, L, seed=7).cuda()
side = torch.cuda.Stream(); side_sp = ctypes.c_void_p(side.cuda_stream); null_sp = ctypes.c_void_p(0)
def fe(sp):
    feat = torch.empty(B, T, 224, device="cuda"); lib.acx_logmel_bn0(h, _ffi.ptr(wav), B, L, _ffi.ptr(feat), 1, sp)
    x = torch.empty(B, 252, 56, 96, device="cuda"); lib.acx_stem_ln(h, _ffi.ptr(feat), B, T, _ffi.ptr(x), sp)
    return feat, x
rf, rx = fe(null_sp); torch.cuda.synchronize()
def load(n):
    if what.startswith("block"):
        s = int(what[5:]); C = (96, 192, 384, 768)[s]; H = (252, 126, 63, 31)[s]; W = (56, 28, 14, 7)[s]
        need = ctypes.c_size_t(); lib.acx_block_scratch_bytes(s, B, H, W, ctypes.byref(need))
        x = torch.randn(B, H, W, C, device="cuda"); scr = torch.empty(need.value, dtype=torch.uint8, device="cuda")
        for _ in range(n): lib.acx_block(h, s, 0, _ffi.ptr(x), B, H, W, _ffi.ptr(scr), need.value, null_sp)
        return x, scr
    if what.startswith("down"):
        i = int(what[4:]); Ci = (96, 192, 384)[i - 1]; Co = (192, 384, 768)[i - 1]; H = (252, 126, 63)[i - 1]; W = (56, 28, 14)[i - 1]
        x = torch.randn(B, H, W, Ci, device="cuda"); out = torch.empty(B, H // 2, W // 2, Co, device="cuda"); scr = torch.empty_like(x)
        for _ in range(n): lib.acx_downsample(h, i, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scr), B, H, W, null_sp)
        return x, out, scr
    return None
bad_f = bad_x = 0
for it in range(20):
    torch.cuda.synchronize()
    keep = load(6)
    with torch.cuda.stream(side):
        f, x = fe(side_sp)
    torch.cuda.synchronize()
    bad_f += int(not torch.equal(f, rf)); bad_x += int(not torch.equal(x, rx))
    if not torch.equal(f, rf) and bad_f <= 2:
        d = (f - rf).abs().reshape(-1, 224)
        fr = (d.amax(dim=1) > 0).nonzero().flatten()
        print("  iter %d: %d frames differ; frame ids (mod 4, first 24): %s" % (it, len(fr), [(int(v), int(v) % 4) for v in fr[:24]]))
        k = int(fr[0]); bins = (d[k] > 0).nonzero().flatten()
        print("   frame %d: %d of 224 bins differ; got %s ref %s" % (k, len(bins), f.reshape(-1, 224)[k, bins[:6]].tolist(), rf.reshape(-1, 224)[k, bins[:6]].tolist()))
        print("   isnan/inf in result:", bool(torch.isnan(f).any()), bool(torch.isinf(f).any()))
print("%s on the null stream: log-mel wrong in %d, stem output wrong in %d of 20 concurrent runs" % (what, bad_f, bad_x))
