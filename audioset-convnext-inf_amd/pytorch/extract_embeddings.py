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
    what: 'logits' -> (527,), 'scene' -> (768,), 'frame' -> (768, T', 7) per clip.  Returns a list (CPU tensors, input order).

    The host side is kept off the critical path (with one clip per launch a forward is ~1.5 ms; torch.stack on a many-core host,
    a pageable copy and a synchronising .cpu() per clip cost ten times that): chunks run largest first, so the model's workspace
    and the pinned staging buffer are sized once instead of growing with every longer clip; a chunk's clips are copied into the
    pinned buffer with plain memcpys and cross PCIe asynchronously; logits / scene rows collect in one device tensor that is
    fetched once at the end (frame embeddings, whose shapes differ, are fetched per chunk)."""
    import numpy as np
    device = next(model.parameters()).device
    fn = {"logits": lambda x: model(x)["clipwise_logits"], "scene": model.forward_scene_embeddings,
          "frame": model.forward_frame_embeddings}[what]
    n = len(waveforms)
    out = [None] * n
    if n == 0:
        return out
    chunks = []
    for length, idx in bucket_by_length([len(w) for w in waveforms]).items():
        for s in range(0, len(idx), max_batch):
            chunks.append((length, idx[s:s + max_batch]))
    chunks.sort(key=lambda c: -c[0] * len(c[1]))                   # stable: equal sizes keep their first-seen order
    pin = torch.empty(chunks[0][0] * len(chunks[0][1]), dtype=torch.float32).pin_memory()
    pin_np = pin.numpy()
    staged = torch.cuda.Event()
    staged.record()
    rows = None                                                    # (n, dim) device tensor of the fixed-size outputs
    for length, chunk in chunks:
        staged.synchronize()                                       # the previous chunk has left the pinned buffer
        view = pin_np[:length * len(chunk)].reshape(len(chunk), length)
        for j, i in enumerate(chunk):
            w = waveforms[i]
            np.copyto(view[j], w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w), casting="same_kind")
        batch = pin[:length * len(chunk)].view(len(chunk), length).to(device, non_blocking=True)
        staged.record()
        res = fn(batch)
        if what == "frame":
            res = res.cpu()                                        # one device -> host copy per chunk, not one per clip
            for j, i in enumerate(chunk):
                out[i] = res[j].clone()
        else:
            if rows is None:
                rows = torch.empty(n, res.shape[1], dtype=res.dtype, device=device)
            rows[torch.as_tensor(chunk, device=device)] = res
    if rows is not None:
        rows = rows.cpu()
        for i in range(n):
            out[i] = rows[i].clone()
    return out
