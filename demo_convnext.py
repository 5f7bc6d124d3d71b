#!/usr/bin/env python
"""Single-clip demo on the MI355X path -- counterpart of the reference's demo_convnext.py (load checkpoint,
load / resample / pad a WAV, three forwards, labels above the 0.25 threshold).

    python demo_convnext.py --ckpt checkpoints/model.safetensors --wav clip.wav --labels metadata/class_labels_indices.csv
    python demo_convnext.py --synthetic-weights --wav clip.wav        # no checkpoint at hand: seeded weights

Prints the same lines as the reference (`# params`, sizes, predicted label indices, names, embedding shapes).
"""
import argparse
import os
import sys

# two hardware queues for this process's HIP streams (an exported value wins): the package's two sub-batch streams and the sweep's
# copy streams overlap best on two -- bench.py and INTEGRATION.md have the measurement.  Before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import numpy as np      # noqa: E402
import torch            # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from audioset_convnext_inf_amd.pytorch.convnext import ConvNeXt, convnext_tiny      # noqa: E402
from audioset_convnext_inf_amd.utils.utilities import default_label_map, read_audioset_label_tags, read_wav_pcm16, prepare_clip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default="topel/ConvNeXt-Tiny-AT", help="local .safetensors/.pth, Zenodo URL or HF model id")
    ap.add_argument("--synthetic-weights", action="store_true", help="seeded synthetic weights instead of a checkpoint")
    ap.add_argument("--wav", required=True)
    ap.add_argument("--labels", default=os.path.join(ROOT, "metadata", "class_labels_indices.csv"))
    ap.add_argument("--threshold", type=float, default=0.25)
    args = ap.parse_args()

    if args.synthetic_weights:
        from audioset_convnext_inf_amd import synth
        model = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                              use_speed_perturb=False)
        model.load_state_dict(synth.synth_state_dict(0))
    else:
        model = ConvNeXt.from_pretrained(args.ckpt, use_auth_token=None, map_location="cpu")
        if model is None:
            sys.exit(1)
    print("# params:", sum(p.numel() for p in model.parameters() if p.requires_grad))
    if not torch.cuda.is_available():
        sys.exit("this build runs on an MI355X; no GPU is visible")
    device = torch.device("cuda")
    model = model.to(device).eval()

    sample_rate = 32000
    print("\nInference on " + os.path.basename(args.wav) + "\n")
    wav, sr = read_wav_pcm16(args.wav)
    waveform = torch.from_numpy(wav[:1])                 # first channel, (1, L)
    if sr != sample_rate:
        print("Resampling from %d to 32000 Hz" % sr)
    if waveform.shape[-1] < 10 * sample_rate and sr == sample_rate:
        print("Padding waveform")
    elif waveform.shape[-1] > 10 * sample_rate and sr == sample_rate:
        print("Cropping waveform")
    waveform = prepare_clip(waveform, sr, sample_rate, 10).to(device)

    with torch.no_grad():
        output = model(waveform)
    logits, probs = output["clipwise_logits"], output["clipwise_output"]
    print("logits size:", logits.size())
    print("probs size:", probs.size())

    sample_labels = np.where(probs[0].clone().detach().cpu() > args.threshold)[0]
    print("Predicted labels using activity threshold %.2f:\n" % args.threshold)
    print(sample_labels)
    # the reference reads metadata/class_labels_indices.csv (demo_convnext.py:29, utilities.py:195-216); without that file the
    # packaged copy of the same table names the classes
    _, ix_to_lb, _, _ = read_audioset_label_tags(args.labels) if os.path.isfile(args.labels) else default_label_map()
    for l in sample_labels:
        print("%s: %.3f" % (ix_to_lb[l], probs[0, l]))

    with torch.no_grad():
        scene = model.forward_scene_embeddings(waveform)
    print("\nScene embedding, shape:", scene.size())
    with torch.no_grad():
        frame = model.forward_frame_embeddings(waveform)
    print("\nFrame-level embeddings, shape:", frame.size())


if __name__ == "__main__":
    main()
