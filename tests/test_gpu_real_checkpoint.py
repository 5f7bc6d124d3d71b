"""GPU, self-arming: the reference's ONLY published known answers, asserted the day a real checkpoint is at hand.

    ACX_CKPT=/path/to/convnext_tiny_471mAP.pth (or model.safetensors)  python -m pytest tests/test_gpu_real_checkpoint.py -m gpu
    ACX_AUDIOSET_EVAL=/path/to/eval   (optional: <prefix>_waveforms.npy + <prefix>_targets.npy, or a packed .h5 / .hdf5 file)

Without ACX_CKPT every test here is SKIPPED with the reason printed (`-rs`); nothing is downloaded.  With it:
  * the checkpoint's frontend buffers pin the restated torchlibrosa / librosa tables (tools/verify_checkpoint.py, exit code 0:
    SURVEY 8c's one strong pin -- `convnext.py:179-200` builds them, the checkpoint stores them);
  * 28 222 767 trainable parameters (/root/reference/README.md:49);
  * the demo clip f62-S-v2swA (its PCM is a committed fixture, tests/golden/g1_demo.npz) gives the label indices
    [0 137 138 139 151 506] at the demo's 0.25 threshold (README.md:57, scripts/demo_convnext.sbatch.output:11), in both
    fp32-grade arithmetics, and output shapes (1, 527) / (1, 768) / (1, 768, 31, 7) (README.md:52-61);
  * with ACX_AUDIOSET_EVAL: mAP within 1e-3 of 0.471 and AUC within 1e-3 of 0.973 over the evaluation set (README.md:34-36),
    through pytorch/evaluate.py exactly as evaluate_convnext_on_audioset.py drives it.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CKPT = os.environ.get("ACX_CKPT", "")
EVAL = os.environ.get("ACX_AUDIOSET_EVAL", "")

pytestmark = [
    pytest.mark.gpu,
    pytest.mark.skipif(not (CKPT and os.path.isfile(CKPT)),
                       reason="known-answer tests of the REAL checkpoint are armed by ACX_CKPT=<convnext_tiny_471mAP.pth | model.safetensors> "
                              "(unset or not a file here: labels [0 137 138 139 151 506], 28 222 767 params, mAP 0.471 stay unchecked)"),
]

DEMO_LABELS = [0, 137, 138, 139, 151, 506]        # /root/reference/README.md:57
N_PARAMS = 28222767                                # /root/reference/README.md:49
MAP, AUC = 0.471, 0.973                            # /root/reference/README.md:34-36


@pytest.fixture(scope="module")
def real_model():
    from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny, load_checkpoint
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
    load_checkpoint(m, CKPT)
    return m.to("cuda").eval()


def test_checkpoint_frontend_buffers_pin_the_restated_tables():
    """The checkpoint carries torchlibrosa's conv_real / conv_imag and librosa's melW: they must equal the tables
    frontend_tables.py restates (<= 1e-6), and libacx must choose the FFT frontend on them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_checkpoint.py"), CKPT], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FFT frontend" in r.stdout


def test_parameter_count(real_model):
    assert sum(p.numel() for p in real_model.parameters() if p.requires_grad) == N_PARAMS


@pytest.mark.parametrize("precision", ["fp32_split", "fp32"])
def test_demo_clip_labels(real_model, precision):
    """demo_convnext.py:47-92 on the reference's own sample: PCM16 / 32768, already 10 s at 32 kHz."""
    d = np.load(os.path.join(GOLD, "g1_demo.npz"))
    wav = torch.from_numpy(d["pcm16"].astype(np.float32) / 32768.0)[None].cuda()
    real_model.set_precision(precision)
    try:
        with torch.no_grad():
            out = real_model(wav)
            scene = real_model.forward_scene_embeddings(wav)
            frame = real_model.forward_frame_embeddings(wav)
    finally:
        real_model.set_precision("fp32_split")
    probs = out["clipwise_output"][0].cpu().numpy()
    labels = np.where(probs > 0.25)[0].tolist()
    print("labels above 0.25:", labels, "probs:", np.round(probs[labels], 3))
    assert labels == DEMO_LABELS
    assert out["clipwise_logits"].shape == (1, 527) and scene.shape == (1, 768) and frame.shape == (1, 768, 31, 7)
    assert frame_info_is_fft(real_model)


def frame_info_is_fft(model):
    info = model.native_context(torch.device("cuda", 0)).frontend_info()
    return not info["dense_dft"]


def test_demo_script_on_the_real_checkpoint(tmp_path):
    """The script a user runs, end to end (WAV reader incl. the LIST chunk, label names)."""
    import wave
    d = np.load(os.path.join(GOLD, "g1_demo.npz"))
    wav_path = str(tmp_path / "f62-S-v2swA_200000_210000.wav")
    with wave.open(wav_path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(32000)
        w.writeframes(d["pcm16"].tobytes())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo_convnext.py"), "--ckpt", CKPT, "--wav", wav_path],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "# params: %d" % N_PARAMS in r.stdout
    assert "[  0 137 138 139 151 506]" in r.stdout


@pytest.mark.skipif(not EVAL, reason="mAP 0.471 / AUC 0.973 need the AudioSet evaluation set: ACX_AUDIOSET_EVAL=<prefix of _waveforms.npy/_targets.npy | .h5>")
def test_audioset_eval_map(real_model):
    from audioset_convnext_inf_amd.pytorch.evaluate import evaluate_sharded
    from audioset_convnext_inf_amd.utils.data_generator import ClipShard
    if EVAL.endswith((".h5", ".hdf5")):
        shard = ClipShard.from_hdf5(EVAL)
    else:
        shard = ClipShard.from_npy(EVAL + "_waveforms.npy", EVAL + "_targets.npy")
    stats = evaluate_sharded(real_model, shard, batch_size=256)
    m_ap, m_auc = float(np.mean(stats["average_precision"])), float(np.mean(stats["auc"]))
    print("mAP %.4f  AUC %.4f over %d clips" % (m_ap, m_auc, len(shard)))
    assert abs(m_ap - MAP) <= 1e-3
    assert abs(m_auc - AUC) <= 1e-3
