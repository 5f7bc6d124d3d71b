"""CPU: the components either side of the hot path: label map, PCM scaling, WAV reading,
evaluation batching and metrics, length bucketing."""
import numpy as np
import pytest
import torch

from audioset_convnext_inf_amd.pytorch import evaluate as ev
from audioset_convnext_inf_amd.pytorch.extract_embeddings import bucket_by_length
from audioset_convnext_inf_amd.utils import utilities as ut
from audioset_convnext_inf_amd.utils.data_generator import ClipShard, EvaluateSampler, evaluate_batches


def test_label_tags(tmp_path):
    p = tmp_path / "labels.csv"
    p.write_text('index,mid,display_name\n0,/m/09x0r,"Speech"\n1,/m/05zppz,"Male speech, man speaking"\n2,/m/04rlf,"Music"\n')
    lb_to_ix, ix_to_lb, id_to_ix, ix_to_id = ut.read_audioset_label_tags(str(p))
    assert ix_to_lb == {0: "Speech", 1: "Male speech, man speaking", 2: "Music"}
    assert lb_to_ix["Music"] == 2 and id_to_ix["/m/09x0r"] == 0 and ix_to_id[1] == "/m/05zppz"


def test_packaged_label_map():
    """The label table that ships with the package (527 AudioSet classes, index order): same four dicts as the CSV reader; the
    reference's published demo answer `[0 137 138 139 151 506]` (README.md:57) names Speech / Music / instruments."""
    lb_to_ix, ix_to_lb, id_to_ix, ix_to_id = ut.default_label_map()
    assert len(ix_to_lb) == 527 and len(ix_to_id) == 527 and len(id_to_ix) == 527
    assert ix_to_lb[0] == "Speech" and ix_to_id[0] == "/m/09x0r" and lb_to_ix["Speech"] == 0 and id_to_ix["/m/09x0r"] == 0
    assert ix_to_lb[137] == "Music"
    assert all(isinstance(v, str) and v for v in ix_to_lb.values())


def test_pcm_scaling_conventions():
    x = np.array([-32768, -1, 0, 1, 32767], dtype=np.int16)
    np.testing.assert_array_equal(ut.int16_to_float32(x), (x / 32767.0).astype(np.float32))      # HDF5 path
    assert ut.float32_to_int16(np.array([2.0, -2.0, 0.5])).tolist() == [32767, -32767, 16383]
    assert ut.pad_or_truncate(np.arange(3), 5).tolist() == [0, 1, 2, 0, 0]
    assert ut.pad_or_truncate(np.arange(7), 5).tolist() == [0, 1, 2, 3, 4]


def test_wav_reader_with_list_chunk(tmp_path):
    rs = np.random.RandomState(0)
    w = (rs.randint(-32768, 32767, size=5000) / 32768.0).astype(np.float32)
    for list_chunk in (False, True):
        p = str(tmp_path / ("a%d.wav" % list_chunk))
        ut.write_wav_pcm16(p, w, 32000, list_chunk=list_chunk)
        got, sr = ut.read_wav_pcm16(p)
        assert sr == 32000 and got.shape == (1, 5000)
        np.testing.assert_array_equal(got[0], w)                    # /32768, the torchaudio.load convention
    clip = ut.prepare_clip(torch.from_numpy(got), 32000)
    assert clip.shape == (1, 320000) and float(clip[0, 5000:].abs().max()) == 0.0
    long = ut.prepare_clip(torch.zeros(1, 400000), 32000)
    assert long.shape == (1, 320000)
    assert ut.prepare_clip(torch.zeros(1, 16000), 16000).shape == (1, 320000)      # resampled then padded


def test_sampler_batches_and_sharding():
    assert [b.tolist() for b in EvaluateSampler(7, 3)] == [[0, 1, 2], [3, 4, 5], [6]]          # short last batch
    a = [b.tolist() for b in EvaluateSampler(10, 3, rank=0, world_size=2)]
    b = [b.tolist() for b in EvaluateSampler(10, 3, rank=1, world_size=2)]
    assert a == [[0, 1, 2], [6, 7, 8]] and b == [[3, 4, 5], [9]]
    assert len(EvaluateSampler(10, 3, 0, 2)) == 2 and len(EvaluateSampler(10, 3, 1, 2)) == 2
    shard = ClipShard(np.arange(40, dtype=np.int16).reshape(5, 8), np.eye(5, 527, dtype=bool))
    batches = list(evaluate_batches(shard, batch_size=2))
    assert [x["waveform"].shape for x in batches] == [(2, 8), (2, 8), (1, 8)]
    assert batches[0]["waveform"].dtype == np.float32 and batches[0]["target"].dtype == np.float32
    np.testing.assert_allclose(batches[2]["waveform"][0], np.arange(32, 40) / 32767.0, rtol=1e-7)


class _Scorer(torch.nn.Module):
    """Deterministic stand-in model: score of class c = sigmoid(mean(x) * (c+1))."""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        z = x.mean(1, keepdim=True) * torch.arange(1, 528, dtype=torch.float32)[None]
        return {"clipwise_output": torch.sigmoid(z), "clipwise_logits": z}


def test_evaluator_statistics_match_sklearn():
    from math import sqrt
    from scipy.stats import norm
    from sklearn import metrics
    rs = np.random.RandomState(1)
    wav = (rs.standard_normal((37, 64)) * 3000).astype(np.int16)
    tgt = rs.uniform(size=(37, 527)) < 0.3
    tgt[0], tgt[1] = True, False
    shard = ClipShard(wav, tgt)
    stats = ev.Evaluator(_Scorer()).evaluate(evaluate_batches(shard, batch_size=8))
    x = torch.from_numpy(wav / np.float32(32767.0)).float()
    scores = _Scorer()(x)["clipwise_output"].numpy()
    np.testing.assert_allclose(stats["average_precision"], metrics.average_precision_score(tgt.astype(np.float32), scores, average=None))
    auc = metrics.roc_auc_score(tgt.astype(np.float32), scores, average=None)
    np.testing.assert_allclose(stats["auc"], auc)
    np.testing.assert_allclose(stats["d_prime"], sqrt(2) * norm.ppf(auc))
    assert stats["average_precision"].shape == (527,)
    # object-dtype batches (the reference's collate_fn output) are cast, not passed through
    obj = np.empty(2, dtype=object)
    obj[0], obj[1] = np.zeros(4), np.ones(4)
    assert ev.move_data_to_device(obj, "cpu").dtype == torch.float32


def test_bucket_by_length():
    assert bucket_by_length([10, 20, 10, 30, 20]) == {10: [0, 2], 20: [1, 4], 30: [3]}


def test_resample_matches_the_interpolation_formula():
    """utils/resample.py restates torchaudio.functional.resample (demo_convnext.py:53-59).  Check it against a direct
    float64 evaluation of the same band-limited interpolation, y[n] = sum_m x[m] h(n / new - m / orig), with
    h(t) = scale * sinc(base t) * cos^2(pi base t / (2 * 6)) on |base t| <= 6, and against an analytic tone."""
    import math
    from audioset_convnext_inf_amd.utils.resample import resample
    rs = np.random.RandomState(3)
    for orig, new in ((44100, 32000), (16000, 32000), (48000, 32000), (8000, 32000)):
        x = rs.standard_normal(3000)
        y = resample(torch.from_numpy(x.astype(np.float32))[None], orig, new)[0].numpy()
        assert y.shape[0] == math.ceil(3000 * new / orig)
        g = math.gcd(orig, new)
        of, nf = orig // g, new // g
        base = min(of, nf) * 0.99
        m = np.arange(3000)
        for n in rs.randint(0, y.shape[0], size=40):
            t = (n / nf - m / of) * base
            h = np.where(np.abs(t) < 6, np.sinc(t) * np.cos(np.clip(t, -6, 6) * math.pi / 12) ** 2, 0.0) * (base / of)
            assert abs(float(np.dot(x, h)) - y[n]) < 2e-5 * max(1.0, np.abs(x).max())
    tone = np.sin(2 * np.pi * 1000 * np.arange(22050) / 44100)
    y = resample(torch.from_numpy(tone.astype(np.float32))[None], 44100, 32000)[0].numpy()
    ref = np.sin(2 * np.pi * 1000 * np.arange(y.shape[0]) / 32000)
    assert np.abs(y - ref)[300:-300].max() < 5e-4
    same = torch.randn(1, 100)
    assert resample(same, 32000, 32000) is same
    clip = ut.prepare_clip(torch.from_numpy(tone.astype(np.float32))[None], 44100)          # demo path: resample, then pad
    assert clip.shape == (1, 320000) and float(clip[0, 16100:].abs().max()) == 0.0


def test_hdf5_shard_reader_with_a_stand_in_h5py(tmp_path, monkeypatch):
    """`ClipShard.from_hdf5` reads the reference's packed format (dataset.py:193-199: audio_name S20, waveform int16
    (N, L), target bool (N, 527), attr sample_rate).  h5py is not in this image: a stand-in module with the slice of
    its API that the reader uses (File as a context manager, dataset[:]) executes the code path."""
    import sys
    import types
    rs = np.random.RandomState(1)
    store = {"waveform": rs.randint(-3000, 3000, size=(5, 640)).astype(np.int16),
             "target": rs.uniform(size=(5, 527)) < 0.2,
             "audio_name": np.array([("Y%04d.wav" % i).encode() for i in range(5)], dtype="S20")}

    class _File(dict):
        def __init__(self, path, mode="r"):
            assert mode == "r" and path.endswith("eval.h5")
            super().__init__(store)
            self.attrs = {"sample_rate": 32000}

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=_File))
    shard = ClipShard.from_hdf5(str(tmp_path / "eval.h5"))
    assert len(shard) == 5 and shard.audio_names.tolist() == ["Y%04d.wav" % i for i in range(5)]
    batches = list(evaluate_batches(shard, batch_size=2))
    assert [b["waveform"].shape[0] for b in batches] == [2, 2, 1]
    np.testing.assert_array_equal(batches[0]["waveform"], (store["waveform"][:2] / 32767.0).astype(np.float32))
    assert batches[2]["target"].dtype == np.float32 and batches[2]["audio_name"][0] == "Y0004.wav"


def test_ahead_reader_thread_stops_when_the_consumer_leaves_early():
    """the background reader of evaluate.forward must not block forever in q.put once its consumer is gone."""
    import threading
    import time
    from audioset_convnext_inf_amd.pytorch import evaluate as ev
    produced = []

    def gen():
        for i in range(1000):
            produced.append(i)
            yield i

    before = threading.active_count()
    it = ev._ahead(gen(), lambda x: x * 2)
    assert next(it) == (0, 0) and next(it) == (1, 2)
    it.close()                                   # consumer leaves (as an exception in the loop body would)
    deadline = time.time() + 5
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() <= before
    assert len(produced) < 10                    # the reader did not run on
    # and an exception in the generator still reaches the consumer
    def bad():
        yield 1
        raise ValueError("boom")
    with pytest.raises(ValueError, match="boom"):
        list(ev._ahead(bad(), lambda x: x))


def test_verify_checkpoint_tool(tmp_path, synth_sd):
    """tools/verify_checkpoint.py (SURVEY 8c's pin for the day a real checkpoint arrives): exit 0 on a checkpoint that carries
    this package's own tables, 1 when a buffer deviates by more than the tolerance, 2 when one is missing -- for both
    checkpoint formats the reference ships (.safetensors, .pth with {"model": sd})."""
    import os
    import subprocess
    import sys
    from safetensors.torch import save_file
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "verify_checkpoint.py")
    good = str(tmp_path / "model.safetensors")
    save_file({k: v.contiguous() for k, v in synth_sd.items()}, good)
    r = subprocess.run([sys.executable, tool, good], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runs the FFT frontend" in r.stdout and "884 taps" in r.stdout and "frontend tables pinned" in r.stdout
    bad = dict(synth_sd)
    bad["logmel_extractor.melW"] = synth_sd["logmel_extractor.melW"] + 1e-4
    pth = str(tmp_path / "bad.pth")
    torch.save({"model": bad}, pth)
    r = subprocess.run([sys.executable, tool, pth], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "DEVIATES" in r.stdout and "frontend tables DIFFER" in r.stdout
    worse = dict(synth_sd)
    worse["spectrogram_extractor.stft.conv_real.weight"] = synth_sd["spectrogram_extractor.stft.conv_real.weight"] * 1.001
    del worse["logmel_extractor.melW"]
    pth2 = str(tmp_path / "worse.pth")
    torch.save(worse, pth2)
    r = subprocess.run([sys.executable, tool, pth2], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "missing buffers" in r.stdout


def test_eight_reader_processes_feed_one_shard(tmp_path):
    """configs[4] on the host side: eight rank processes, each staging ITS batches of one memory-mapped
    shard exactly as pytorch/evaluate.py does (8 copy threads per process), must together sustain what eight GPUs consume --
    8 x 5.2 GB/s at 8 000 clips/s per rank -- on a host that has the cores for it (>= 32; a 256-CPU node feeds 8 MI355X).
    The 8-core build container is asked for half of that per process; the figures are printed either way."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shard = str(tmp_path / "eval_waveforms.npy")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "feed_bench.py"), "--make", "1024", "--shard", shard, "--procs", "8",
                        "--batch", "64", "--seconds", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    print("reader processes:", rec)
    assert rec["procs"] == 8 and all(g > 0 for g in rec["GBs_per_proc"])            # every rank got batches
    need = 8 * 5.2 if (os.cpu_count() or 1) >= 32 else 8 * 2.6
    assert rec["GBs_total"] >= need, "8 reader processes stage %.1f GB/s, the sweep on 8 GPUs needs %.1f" % (rec["GBs_total"], need)
