#!/usr/bin/env python
"""Golden vector g4_stress.npz: the REFERENCE model class under the trained-like stress weights
(`synth.stress_state_dict(0)`: LayerNorm weights over > 3 decades, gamma over 5 decades, outlier hidden units,
|mean| / std of the LayerNorm input near 20 -- tests/test_gpu_stress.py).

Same procedure and stand-ins as make_goldens.py (runs ONLY in the build container, imports the reference from
/root/reference/src, copies nothing): the reference's own forward / forward_scene_embeddings /
forward_frame_embeddings on two 1.25 s clips.  Pins the oracle -- and through it the HIP path -- to the reference under
the weight statistics the default arithmetic was questioned on, not only under the homogeneous synthetic weights.

usage: python tests/golden/make_stress_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_goldens as mg                              # noqa: E402  (stand-ins for torchlibrosa / torchaudio)
from oracle import ref_cpu                             # noqa: E402
from audioset_convnext_inf_amd import synth            # noqa: E402


def main():
    mg._install_shims()
    sys.path.insert(0, os.path.join(mg.REF, "src"))
    from audioset_convnext_inf.pytorch.convnext import convnext_tiny   # the reference itself

    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                          use_speed_perturb=False)
    sd = synth.stress_state_dict(0)
    model.load_state_dict(sd, strict=True)
    model.eval()
    wav = torch.cat([synth.synth_waveforms(1, 40000, seed=21), 0.3 * synth.synth_waveforms(1, 40000, seed=22)])
    with torch.no_grad():
        o = model(wav)
        ref = {"logits": o["clipwise_logits"], "probs": o["clipwise_output"],
               "scene": model.forward_scene_embeddings(wav), "frame": model.forward_frame_embeddings(wav)}
    oo = ref_cpu.forward(sd, wav)
    orc = {"logits": oo["clipwise_logits"], "probs": oo["clipwise_output"],
           "scene": ref_cpu.forward_scene_embeddings(sd, wav), "frame": ref_cpu.forward_frame_embeddings(sd, wav)}
    dev = {k: float((ref[k].double() - orc[k].double()).abs().max()) for k in ref}
    mag = {k: float(ref[k].double().abs().max()) for k in ref}
    print("oracle vs reference under stress weights: max abs dev", dev, "max |ref|", mag)
    np.savez_compressed(os.path.join(HERE, "g4_stress.npz"), wav=wav.numpy(), **{k: v.numpy() for k, v in ref.items()})
    with open(os.path.join(HERE, "MANIFEST_stress.json"), "w") as f:
        json.dump({"weights": "synth.stress_state_dict(0)", "weights_sha256": synth.state_dict_digest(sd),
                   "oracle_vs_reference": dev, "max_abs_reference": mag,
                   "shapes": {k: list(v.shape) for k, v in ref.items()}}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
