"""Instrumented FFT victim (tools/race2/fe_probe.hip) on a side stream next to libacx kernels on the null stream.
    ACX_LIB=build/variants/libacx_xxx.so python tools/race2/run_probe.py down2|block0|block1|block2|block3|none [iters]
Prints which probe kinds fired (see fe_probe.hip) and the first events."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
what = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = os.environ.get("ACX_PRECISION", "fp32_split")
pr = ctypes.CDLL(os.path.join(ROOT, "build", "variants", "libfeprobe.so"))
pr.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
L = 320000; B = 32; T = L // 320 + 1; NF = B * T
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
assert pr.probe_init() == 0
wav = synth.synth_waveforms(B, L, seed=7).cuda()
side = torch.cuda.Stream(); side_sp = ctypes.c_void_p(side.cuda_stream); null_sp = ctypes.c_void_p(0)
NREP = 16 + 256 * 16
def victim(sp):
    power = torch.empty(NF, 512, device="cuda"); rep = torch.zeros(NREP, dtype=torch.int32, device="cuda")
    assert pr.probe_launch(wav.data_ptr(), L, B, power.data_ptr(), rep.data_ptr(), sp) == 0
    return power, rep
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
pr.valu_victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
seed = torch.randn(65536, device="cuda"); VB = 4096
def valu(sp):
    o = torch.empty(VB * 256, device="cuda")
    assert pr.valu_victim_launch(seed.data_ptr(), o.data_ptr(), VB, 400, sp) == 0
    return o
vref = valu(null_sp); vbad = vbad_runs = 0
ref, rep0 = victim(null_sp); torch.cuda.synchronize()
assert int(rep0[:16].sum()) == 0, "probe fired without an aggressor: %s" % rep0[:16].tolist()
tot = [0] * 16; bad_runs = 0; shown = 0
for it in range(iters):
    torch.cuda.synchronize()
    keep = load(8)
    with torch.cuda.stream(side):
        p, rep = victim(side_sp)
    torch.cuda.synchronize()
    keep = load(8)
    with torch.cuda.stream(side):
        vo = valu(side_sp)
    torch.cuda.synchronize()
    nb = int((vo != vref).sum()); vbad += nb; vbad_runs += int(nb > 0)
    r = rep.cpu().numpy().astype("uint32")
    wrong = int((p != ref).any(dim=1).sum())
    bad_runs += int(wrong > 0)
    for k in range(16): tot[k] += int(r[k])
    print("iter %d: frames with wrong power spectrum %d; probe counts kind1..8 = %s" % (it, wrong, [int(r[k]) for k in range(1, 9)]))
    n = min(int(r[15]), 256)
    for e in range(n):
        if shown >= 24: break
        ev = r[16 + 16 * e: 32 + 16 * e]
        import struct
        f = lambda u: struct.unpack("f", struct.pack("I", int(u)))[0]
        print("   kind %d wg %d tid %d frame %d slot %d  found (%.6g, %.6g) expected (%.6g, %.6g) heal %d  hw_id 0x%08x xcc %d lds_alloc 0x%08x"
              % (ev[0], ev[1], ev[2], ev[3], ev[4], f(ev[5]), f(ev[7]), f(ev[6]), f(ev[8]), ev[9], ev[10], ev[11], ev[12]))
        shown += 1
print("SUMMARY lib=%s prec=%s aggressor=%s: wrong in %d of %d runs; totals kind1..8 = %s; pure-VALU victim wrong in %d runs (%d threads)"
      % (os.path.basename(os.environ.get("ACX_LIB", "libacx.so")), prec, what, bad_runs, iters, tot[1:9], vbad_runs, vbad))
