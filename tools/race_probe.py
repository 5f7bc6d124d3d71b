"""Localise run-to-run differences of the two-stream forward: which clips / rows deviate from the one-stream result."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny

def make(split):
    os.environ["ACX_SPLIT_STREAMS"] = "1" if split else "0"
    m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0))
    m = m.to("cuda").eval()
    m(synth.synth_waveforms(1, 32000, seed=1).cuda())     # creates the context under this env
    return m
wav = synth.synth_waveforms(64, 320000, seed=1234).cuda()
one = make(False); two = make(True)
ref = one.forward_frame_embeddings(wav).clone(); torch.cuda.synchronize()
ref2 = one.forward_frame_embeddings(wav).clone(); torch.cuda.synchronize()
print("one-stream run-to-run identical:", torch.equal(ref, ref2))
fill = int(os.environ.get("PROBE_FILL", "-1"))
for it in range(6):
    if fill >= 0 and two._ws:
        for w in two._ws.values():
            w.fill_(fill)                     # poison the workspace: 255 -> NaN patterns everywhere
        torch.cuda.synchronize()
    out = two.forward_frame_embeddings(wav); torch.cuda.synchronize()
    d = (out - ref).abs()
    bad = (d.amax(dim=(1, 2, 3)) > 0).nonzero().flatten().tolist()
    msg = "iter %d: clips differing from the one-stream result: %s" % (it, bad)
    for b in bad[:3]:
        pos = (d[b].amax(dim=0) > 0).nonzero()
        msg += " | clip %d: %d of 217 positions, h range %d..%d, max %.3g" % (b, len(pos), int(pos[:, 0].min()), int(pos[:, 0].max()), float(d[b].max()))
    print(msg)
