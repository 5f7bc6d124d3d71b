"""I/O helpers either side of the hot path ("next" rows of SURVEY.md 8f): label map, PCM scaling, WAV
reading, clip padding.  Counterparts of reference `utils/utilities.py` (read_audioset_label_tags :195-216,
int16_to_float32 :226-227, float32_to_int16 :220-223, pad_or_truncate) and of the loading / padding lines of
`demo_convnext.py:52-67` (torchaudio.load convention: PCM16 / 32768)."""
import csv
import json
import os
import struct

import numpy as np


def read_audioset_label_tags(class_labels_indices_csv):
    """metadata/class_labels_indices.csv (index,mid,display_name; 527 rows + header) ->
    (lb_to_ix, ix_to_lb, id_to_ix, ix_to_id), as the reference returns them."""
    with open(class_labels_indices_csv, "r") as f:
        lines = list(csv.reader(f, delimiter=","))
    ids = [row[1] for row in lines[1:]]
    labels = [row[2] for row in lines[1:]]
    lb_to_ix = {label: i for i, label in enumerate(labels)}
    ix_to_lb = {i: label for i, label in enumerate(labels)}
    id_to_ix = {mid: i for i, mid in enumerate(ids)}
    ix_to_id = {i: mid for i, mid in enumerate(ids)}
    return lb_to_ix, ix_to_lb, id_to_ix, ix_to_id


def default_label_map():
    """The same four dicts from the label table that ships with the package (metadata/audioset_class_labels.json: AudioSet's 527
    classes in index order, generated from the reference's metadata/class_labels_indices.csv by tools/make_label_map.py) -- for
    callers that have no CSV at hand (demo_convnext.py without --labels)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metadata", "audioset_class_labels.json")
    with open(path, "r") as f:
        classes = json.load(f)["classes"]
    ids = [c[0] for c in classes]
    labels = [c[1] for c in classes]
    return ({label: i for i, label in enumerate(labels)}, {i: label for i, label in enumerate(labels)},
            {mid: i for i, mid in enumerate(ids)}, {i: mid for i, mid in enumerate(ids)})


def float32_to_int16(x):
    return (np.clip(x, -1, 1) * 32767.0).astype(np.int16)


def int16_to_float32(x):
    """HDF5 waveforms are stored as int16 and scaled by 1/32767 (NOT 32768) -- utilities.py:226-227."""
    return (x / 32767.0).astype(np.float32)


def pad_or_truncate(x, audio_length):
    """Right zero-pad or crop a 1-D array to `audio_length` samples."""
    if len(x) <= audio_length:
        return np.concatenate((x, np.zeros(audio_length - len(x), dtype=x.dtype)), axis=0)
    return x[0:audio_length]


def read_wav_pcm16(path):
    """Minimal RIFF/WAVE reader for mono/stereo PCM16 (tolerates extra chunks such as LIST before `data`,
    which the reference's sample clip has).  Returns (float32 array (channels, samples) scaled by 1/32768 --
    the torchaudio.load convention used at demo_convnext.py:52 -- and the sample rate)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise ValueError("%s is not a RIFF/WAVE file" % path)
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid = buf[pos:pos + 4]
        size = struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        body = buf[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("%s: missing fmt or data chunk" % path)
    audio_format, channels, sample_rate, _, _, bits = fmt
    if audio_format != 1 or bits != 16:
        raise ValueError("%s: only PCM16 is supported (format %d, %d bits)" % (path, audio_format, bits))
    pcm = np.frombuffer(data, dtype="<i2")
    pcm = pcm[: (len(pcm) // channels) * channels].reshape(-1, channels).T
    return (pcm.astype(np.float32) / 32768.0), sample_rate


def write_wav_pcm16(path, waveform, sample_rate=32000, list_chunk=False):
    """Test helper: write float waveform (samples,) or (channels, samples) as PCM16."""
    w = np.atleast_2d(np.asarray(waveform, dtype=np.float32))
    pcm = np.clip(np.round(w * 32768.0), -32768, 32767).astype("<i2").T.copy()
    channels = pcm.shape[1]
    data = pcm.tobytes()
    fmt = struct.pack("<HHIIHH", 1, channels, sample_rate, sample_rate * channels * 2, channels * 2, 16)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if list_chunk:
        body = b"INFOISFT" + struct.pack("<I", 6) + b"acx\x00\x00\x00"
        chunks += b"LIST" + struct.pack("<I", len(body)) + body
    chunks += b"data" + struct.pack("<I", len(data)) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def prepare_clip(waveform, sample_rate, target_rate=32000, target_seconds=10):
    """demo_convnext.py:53-67: (resample if needed,) right zero-pad / crop to 10 s.  waveform: torch (1, L)."""
    import torch
    import torch.nn.functional as F
    if sample_rate != target_rate:
        # torchaudio.functional.resample semantics (windowed sinc, width 6, rolloff 0.99): utils/resample.py
        from .resample import resample
        waveform = resample(waveform.to(torch.float32), int(sample_rate), int(target_rate))
    n = target_rate * target_seconds
    if waveform.shape[-1] < n:
        waveform = F.pad(waveform, (0, n - waveform.shape[-1]), mode="constant", value=0.0)
    elif waveform.shape[-1] > n:
        waveform = waveform[:, :n]
    return waveform.contiguous()
