"""Soak run (GPU box): the same forward 150-300 times per precision and shape; every result must equal the first bit for bit
(catches schedule-dependent defects -- missing waits, ring races -- that a single parity run can pass by luck)."""
import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
sd = synth.synth_state_dict(0)
for prec in ("fp32_split", "bf16", "bf16a", "fp32"):
    m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(sd); m = m.to("cuda").eval().set_precision(prec)
    for B, L in ((64, 320000), (5, 52000), (32, 160000)):
        wav = synth.synth_waveforms(B, L, seed=B).cuda()
        ref = m(wav)["clipwise_logits"].clone(); reff = m.forward_frame_embeddings(wav).clone()
        bad = 0
        n = 150 if B == 64 else 300
        for i in range(n):
            o = m(wav)["clipwise_logits"]
            if i % 10 == 0:
                f = m.forward_frame_embeddings(wav)
                bad += int(not torch.equal(f, reff))
            bad += int(not torch.equal(o, ref))
        torch.cuda.synchronize()
        print(prec, B, L, "mismatches in %d runs:" % n, bad, "finite:", bool(torch.isfinite(ref).all()))
