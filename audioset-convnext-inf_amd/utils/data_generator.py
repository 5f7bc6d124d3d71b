"""Evaluation-side data plumbing ("next" row 1 of SURVEY.md 8f).  Counterpart of the evaluation parts of
reference `utils/data_generator.py` (AudioSetDataset :27-123, EvaluateSampler :451-501, collate_fn :504-526):
sequential batches (short last batch), int16 -> float32 by /32767, targets kept next to the clips.

Storage: the reference packs AudioSet into HDF5 (`waveform` int16 (N,320000), `target` bool (N,527),
`audio_name` S20; dataset.py:193-199).  h5py is optional here; a raw .npy shard pair (int16 waveforms +
bool/uint8 targets, memory-mapped) is the native format, and `from_hdf5` reads the reference's files when
h5py is importable.

Unlike the reference's collate_fn (object-dtype arrays that reach the model un-cast) batches are
plain float32 (B, L) arrays.
"""
import numpy as np

from .utilities import int16_to_float32


class ClipShard(object):
    """N clips of equal length as int16 + multi-hot targets."""

    def __init__(self, waveforms_int16, targets, audio_names=None):
        assert waveforms_int16.ndim == 2 and waveforms_int16.dtype == np.int16
        assert targets.shape[0] == waveforms_int16.shape[0]
        self.waveforms = waveforms_int16
        self.targets = targets
        self.audio_names = audio_names if audio_names is not None else np.array(
            ["clip_%07d" % i for i in range(waveforms_int16.shape[0])])

    @classmethod
    def from_npy(cls, waveform_npy, target_npy, mmap=True):
        w = np.load(waveform_npy, mmap_mode="r" if mmap else None)
        t = np.load(target_npy, mmap_mode="r" if mmap else None)
        return cls(w, t)

    @classmethod
    def from_hdf5(cls, hdf5_path):
        import h5py          # optional dependency (the reference's packed format)
        with h5py.File(hdf5_path, "r") as hf:
            return cls(hf["waveform"][:], hf["target"][:], np.array([n.decode() for n in hf["audio_name"][:]]))

    def __len__(self):
        return self.waveforms.shape[0]


class EvaluateSampler(object):
    """Sequential index batches; the last one may be short (data_generator.py:478-501)."""

    def __init__(self, audios_num, batch_size, rank=0, world_size=1):
        self.audios_num, self.batch_size = int(audios_num), int(batch_size)
        self.rank, self.world_size = rank, world_size

    def __iter__(self):
        pointer, n = 0, 0
        while pointer < self.audios_num:
            idx = np.arange(pointer, min(pointer + self.batch_size, self.audios_num))
            if n % self.world_size == self.rank:       # rank r takes batches r, r+W, ...
                yield idx
            pointer += self.batch_size
            n += 1

    def __len__(self):
        total = (self.audios_num + self.batch_size - 1) // self.batch_size
        return (total - self.rank + self.world_size - 1) // self.world_size


def evaluate_batches(shard, batch_size=256, rank=0, world_size=1, device_cast=False):
    """Generator of {'audio_name', 'waveform' float32 (B,L), 'target' float32 (B,527)} dicts.

    device_cast=True leaves the clips as they are stored -- 'waveform' is then the int16 (B,L) array -- and
    pytorch/evaluate.py::forward does the /32767 on the GPU (the same double-precision division and float32 rounding as
    utilities.int16_to_float32: the scores are bit-identical), so that half the bytes cross PCIe and the host does no
    arithmetic on 82 M samples per batch of 256."""
    for idx in EvaluateSampler(len(shard), batch_size, rank, world_size):
        # the sampler's batches are runs of consecutive clips: a slice is a view (no copy of 164 MB per batch of 256; for a
        # memory-mapped shard the pages are read when the batch is staged)
        w = shard.waveforms[int(idx[0]):int(idx[-1]) + 1] if device_cast else np.asarray(shard.waveforms[idx])
        yield {
            "audio_name": shard.audio_names[idx],
            "waveform": w if device_cast else int16_to_float32(w),
            "target": np.asarray(shard.targets[idx]).astype(np.float32),
        }
