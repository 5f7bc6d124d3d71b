"""Stress one ConvNeXt block (acx_block, split arithmetic) on two streams at once against its serial result."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0))
C = (96, 192, 384, 768)[stage]; H = (252, 126, 63, 31)[stage]; W = (56, 28, 14, 7)[stage]
need = ctypes.c_size_t(); _ffi.check(_ffi.lib().acx_block_scratch_bytes(stage, B, H, W, ctypes.byref(need)))
torch.manual_seed(0)
xs = [torch.randn(B, H, W, C, device="cuda") for _ in range(2)]
scr = [torch.empty(need.value, dtype=torch.uint8, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(i, stream):
    x = xs[i].clone()
    _ffi.check(_ffi.lib().acx_block(ctx.handle, stage, 0, _ffi.ptr(x), B, H, W, _ffi.ptr(scr[i]), need.value,
                                    ctypes.c_void_p(stream.cuda_stream)))
    return x
refs = []
for i in range(2):
    with torch.cuda.stream(streams[0]):
        refs.append(run(i, streams[0]))
    torch.cuda.synchronize()
bad = 0
for it in range(40):
    outs = []
    torch.cuda.synchronize()
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            outs.append(run(i, streams[i]))
    torch.cuda.synchronize()
    for i in range(2):
        if not torch.equal(outs[i], refs[i]):
            d = (outs[i] - refs[i]).abs()
            rows = (d.reshape(-1, C).amax(dim=1) > 0).nonzero().flatten()
            print("iter %d stream %d: %d rows differ (first %d last %d of %d), max %.3g" % (it, i, len(rows), int(rows[0]), int(rows[-1]), B * H * W, float(d.max())))
            bad += 1
print("stage %d B %d: %d mismatching runs of 80" % (stage, B, bad))
