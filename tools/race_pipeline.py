"""Replays the forward kernel by kernel (per-kernel C ABI) as two half batches on two streams -- the null stream and a
side stream, like acx_forward -- with fresh buffers each time, and reports the first stage whose result differs from
the serial replay."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
L = 320000; B = 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
DIMS = (96, 192, 384, 768); DEPTHS = (3, 3, 9, 3)
T = L // 320 + 1
HW = [_ffi.stage_hw(L, s) for s in range(4)]
def replay(wav, sp, taps):
    feat = torch.empty(B, T, 224, device="cuda"); lib.acx_logmel_bn0(h, _ffi.ptr(wav), B, L, _ffi.ptr(feat), 1, sp)
    x = torch.empty(B, HW[0][0], HW[0][1], 96, device="cuda"); lib.acx_stem_ln(h, _ffi.ptr(feat), B, T, _ffi.ptr(x), sp)
    taps.append(("stem", x.clone()))
    for s in range(4):
        Hs, Ws = HW[s]
        if s > 0:
            Hp, Wp = HW[s - 1]
            nx = torch.empty(B, Hs, Ws, DIMS[s], device="cuda"); scr = torch.empty(B, Hp, Wp, DIMS[s - 1], device="cuda")
            _ffi.check(lib.acx_downsample(h, s, _ffi.ptr(x), _ffi.ptr(nx), _ffi.ptr(scr), B, Hp, Wp, sp)); x = nx
            taps.append(("ds%d" % s, x.clone()))
        need = ctypes.c_size_t(); lib.acx_block_scratch_bytes(s, B, Hs, Ws, ctypes.byref(need))
        scr = torch.empty(need.value, dtype=torch.uint8, device="cuda")
        for j in range(DEPTHS[s]):
            _ffi.check(lib.acx_block(h, s, j, _ffi.ptr(x), B, Hs, Ws, _ffi.ptr(scr), need.value, sp))
            if j == 0 or j == DEPTHS[s] - 1:
                taps.append(("s%d.b%d" % (s, j), x.clone()))
    return x
wavs = [synth.synth_waveforms(B, L, seed=100 + i).cuda() for i in range(2)]
side = torch.cuda.Stream()
null_sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); side_sp = ctypes.c_void_p(side.cuda_stream)
# serial references (final tensors only; taps alias x so only the final state is kept)
refs = []; reftaps = []
for i in range(2):
    t = []; refs.append(replay(wavs[i], null_sp, t).clone()); reftaps.append(t); torch.cuda.synchronize()
for it in range(4):
    torch.cuda.synchronize()
    t0, t1 = [], []
    o0 = replay(wavs[0], null_sp, t0)
    with torch.cuda.stream(side):
        o1 = replay(wavs[1], side_sp, t1)
    torch.cuda.synchronize()
    for i, tt in enumerate((t0, t1)):
        for (name, a), (_, r) in zip(tt, reftaps[i]):
            if not torch.equal(a, r):
                d = (a - r).abs(); rows = (d.reshape(-1, d.shape[-1]).amax(dim=1) > 0).nonzero().flatten()
                print("   iter %d half %d: first deviating tap %s: %d of %d rows, first %d last %d, max %.3g" % (it, i, name, len(rows), d.numel() // d.shape[-1], int(rows[0]), int(rows[-1]), float(d.max())))
                break
    for i, o in enumerate((o0, o1)):
        if not torch.equal(o, refs[i]):
            d = (o - refs[i]).abs().reshape(B, -1).amax(dim=1)
            print("iter %d half %d (%s stream): clips %s differ, max %.3g" % (it, i, "null" if i == 0 else "side", (d > 0).nonzero().flatten().tolist(), float(d.max())))
print("done")
