"""Stress acx_downsample (split arithmetic) on two streams at once against its serial result."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
i_ds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0))
Ci = (96, 192, 384)[i_ds - 1]; Co = (192, 384, 768)[i_ds - 1]; H = (252, 126, 63)[i_ds - 1]; W = (56, 28, 14)[i_ds - 1]
torch.manual_seed(0)
xs = [torch.randn(B, H, W, Ci, device="cuda") for _ in range(2)]
scr = [torch.empty_like(xs[0]) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(i, stream):
    out = torch.empty(B, H // 2, W // 2, Co, device="cuda")
    _ffi.check(_ffi.lib().acx_downsample(ctx.handle, i_ds, _ffi.ptr(xs[i]), _ffi.ptr(out), _ffi.ptr(scr[i]), B, H, W,
                                         ctypes.c_void_p(stream.cuda_stream)))
    return out
refs = []
for i in range(2):
    with torch.cuda.stream(streams[0]):
        refs.append(run(i, streams[0]))
    torch.cuda.synchronize()
bad = 0
for it in range(40):
    outs = []
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            outs.append(run(i, streams[i]))
    torch.cuda.synchronize()
    for i in range(2):
        if not torch.equal(outs[i], refs[i]):
            bad += 1
            d = (outs[i] - refs[i]).abs(); rows = (d.reshape(-1, Co).amax(dim=1) > 0).nonzero().flatten()
            print("iter %d stream %d: %d rows differ (first %d last %d of %d) max %.3g" % (it, i, len(rows), int(rows[0]), int(rows[-1]), d.numel() // Co, float(d.max())))
print("downsample %d: %d mismatching runs of 80" % (i_ds, bad))
