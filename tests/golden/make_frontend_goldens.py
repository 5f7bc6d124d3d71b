#!/usr/bin/env python
"""Golden vectors of the FRONTEND PROBE family: the reference class's outputs on signals whose spectrum has
bins far under the frame peak -- tones at and between bin centres, a 10 s chirp, an impulse train, DC + noise, a full-scale click
over digital silence, a loud burst over near-silence (audioset_convnext_inf_amd.synth.FRONTEND_PROBES).

Build container only (imports the reference from /root/reference through make_goldens.py's stand-ins for torchlibrosa / torchaudio);
stores per probe: sha256 of the waveform (the test regenerates it from the same recipe), logmel (the reference's
`logmel_extractor` output), logits, probs, scene, frame -> tests/golden/g5_frontend.npz, and the oracle's deviation from the
reference -> MANIFEST.json["oracle_vs_reference"]["g5_frontend/<probe>"].

usage: python tests/golden/make_frontend_goldens.py
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_goldens as mg                                  # noqa: E402  (the stand-ins and nothing else)
from oracle import ref_cpu                                 # noqa: E402
from audioset_convnext_inf_amd import synth               # noqa: E402


def main():
    mg._install_shims()
    sys.path.insert(0, os.path.join(mg.REF, "src"))
    from audioset_convnext_inf.pytorch.convnext import convnext_tiny   # the reference itself

    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
    sd = synth.synth_state_dict(0)
    model.load_state_dict(sd, strict=True)
    model.eval()

    out, dev = {}, {}
    for name, L in synth.FRONTEND_PROBES:
        wav = synth.frontend_probe(name)
        assert wav.shape == (1, L)
        taps = {}
        h = model.logmel_extractor.register_forward_hook(lambda m, i, r: taps.__setitem__("logmel", r.detach().clone()))
        with torch.no_grad():
            o = model(wav)
            h.remove()
            r = {"logmel": taps["logmel"], "logits": o["clipwise_logits"], "probs": o["clipwise_output"],
                 "scene": model.forward_scene_embeddings(wav), "frame": model.forward_frame_embeddings(wav)}
        t = {}
        oo = ref_cpu.forward(sd, wav, taps=t)
        orc = {"logmel": t["logmel"], "logits": oo["clipwise_logits"],
               "probs": oo["clipwise_output"], "scene": ref_cpu.forward_scene_embeddings(sd, wav),
               "frame": ref_cpu.forward_frame_embeddings(sd, wav)}
        d = {k: float((r[k].double() - orc[k].double().reshape(r[k].shape)).abs().max()) for k in r}
        dev["g5_frontend/" + name] = d
        print("%-20s logmel %7.2f .. %6.2f dB | oracle vs reference: %s" % (
            name, float(r["logmel"].min()), float(r["logmel"].max()), " ".join("%s %.1e" % kv for kv in d.items())))
        assert max(d.values()) <= 1e-4, d
        out[name + "/sha256"] = np.frombuffer(hashlib.sha256(wav.numpy().tobytes()).digest(), dtype=np.uint8)
        for k, v in r.items():
            out[name + "/" + k] = v.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g5_frontend.npz"), **out)
    mpath = os.path.join(HERE, "MANIFEST.json")
    with open(mpath) as f:
        m = json.load(f)
    m["oracle_vs_reference"].update(dev)
    m["shapes"]["g5_frontend"] = {k: list(v.shape) for k, v in out.items() if not k.endswith("sha256")}
    with open(mpath, "w") as f:
        json.dump(m, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "g5_frontend.npz"))


if __name__ == "__main__":
    main()
