#!/usr/bin/env python
"""Counterpart of the reference's convert_pytorch_ckpt_to_safetensors.py:1-21: load a training checkpoint
(`.pth` holding {"model": state_dict}, or a bare state_dict, or an existing .safetensors) through
`ConvNeXt.from_pretrained`, print the trainable parameter count and write `model.safetensors` (the 190-key file
`ConvNeXt.from_pretrained` / `safetensors.torch.load_model` read back, convnext.py:507).

    python convert_pytorch_ckpt_to_safetensors.py --ckpt checkpoints/convnext_tiny_471mAP.pth [--out model.safetensors]

The reference hard-codes its cluster path; here the paths are arguments.  Pure host-side I/O: no GPU involved."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from audioset_convnext_inf_amd.pytorch.convnext import ConvNeXt      # noqa: E402


def convert(ckpt_path, out_path="model.safetensors"):
    from safetensors.torch import save_model
    model = ConvNeXt.from_pretrained(ckpt_path, use_auth_token=None, map_location="cpu")
    if model is None:
        raise SystemExit("could not load %s" % ckpt_path)
    print("# params:", sum(param.numel() for param in model.parameters() if param.requires_grad))
    save_model(model, out_path)
    return out_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", required=True, help=".pth ({'model': state_dict}) or .safetensors checkpoint")
    ap.add_argument("--out", default="model.safetensors")
    a = ap.parse_args()
    convert(a.ckpt, a.out)
