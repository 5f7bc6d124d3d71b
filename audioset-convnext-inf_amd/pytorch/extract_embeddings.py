"""Variable-length extraction ("next" row 4 of SURVEY.md 8f).  The reference's `pytorch/extract_embeddings.py`
(:64-99) feeds un-padded clips of arbitrary length one at a time (bs=1).  The model is fully convolutional, so
clips of EQUAL length can share a launch: this helper buckets clips by length, runs each bucket as one batch
(chunked to `max_batch`) and returns results in the original order.  Output length follows
T' = ((L//320 + 1) + 4)//4 + 1 -> //2 -> //2 -> //2."""
import torch


def bucket_by_length(lengths):
    """-> {length: [indices]} preserving first-seen order of lengths."""
    buckets = {}
    for i, n in enumerate(lengths):
        buckets.setdefault(int(n), []).append(i)
    return buckets


@torch.no_grad()
def extract(model, waveforms, what="logits", max_batch=64):
    """waveforms: list of 1-D float tensors/arrays of arbitrary lengths (>= 7360 samples).
    what: 'logits' -> (527,), 'scene' -> (768,), 'frame' -> (768, T', 7) per clip.  Returns a list."""
    device = next(model.parameters()).device
    fn = {"logits": lambda x: model(x)["clipwise_logits"], "scene": model.forward_scene_embeddings,
          "frame": model.forward_frame_embeddings}[what]
    out = [None] * len(waveforms)
    for length, idx in bucket_by_length([len(w) for w in waveforms]).items():
        for s in range(0, len(idx), max_batch):
            chunk = idx[s:s + max_batch]
            batch = torch.stack([torch.as_tensor(waveforms[i], dtype=torch.float32) for i in chunk]).to(device)
            res = fn(batch).cpu()            # one device -> host copy per chunk, not one per clip
            for j, i in enumerate(chunk):
                out[i] = res[j].clone()
    return out
