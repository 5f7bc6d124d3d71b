"""GPU: frontends that are NOT torchlibrosa's tables (missing item 2).  The reference applies whatever
`spectrogram_extractor.stft.conv_real/conv_imag.weight` and `logmel_extractor.melW` its state_dict holds
(convnext.py:179-200, 298-299; load_state_dict overwrites the constructor's values): so does the HIP path --
window x DFT buffers of ANY window run on the FFT kernel, anything else as the dense contraction it is (GEMM on the f32
matrix cores), any melW (banded or dense) through the same filter loop.  The oracle always multiplies the stored weights
(oracle/ref_cpu.py:26-47)."""
import numpy as np
import pytest
import torch

from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

KR, KI, KM = ("spectrogram_extractor.stft.conv_real.weight", "spectrogram_extractor.stft.conv_imag.weight",
              "logmel_extractor.melW")


def make(sd, precision="fp32_split"):
    m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
    m.load_state_dict(sd)
    return m.to("cuda").eval().set_precision(precision)


def variants(base):
    g = torch.Generator().manual_seed(11)
    n = torch.arange(1024, dtype=torch.float64)
    hamming = 0.54 - 0.46 * torch.cos(2 * np.pi * n / 1024)
    hann = 0.5 - 0.5 * torch.cos(2 * np.pi * n / 1024)
    out = {}
    sd = dict(base)                                        # another window: still window x DFT -> FFT path
    sd[KR] = (base[KR].double() / hann.clamp_min(1e-30) * hamming).float()
    sd[KI] = (base[KI].double() / hann.clamp_min(1e-30) * hamming).float()
    sd[KR][:, :, 0] = (hamming[0] * 1.0).float()           # (hann[0] = 0: rebuild sample 0 -- cos(0) = 1, sin(0) = 0)
    sd[KI][:, :, 0] = 0.0
    out["hamming_window"] = (sd, False)
    sd = dict(base)                                        # not separable into window x DFT: the dense contraction
    sd[KR] = base[KR] + 2e-3 * torch.randn(base[KR].shape, generator=g)
    sd[KI] = base[KI] + 2e-3 * torch.randn(base[KI].shape, generator=g)
    out["perturbed_stft"] = (sd, True)
    sd = dict(base)                                        # dense (non-banded) mel matrix, FFT path
    sd[KM] = base[KM] + 1e-4 * torch.rand(base[KM].shape, generator=g)
    out["dense_melW"] = (sd, False)
    sd = dict(out["perturbed_stft"][0])                    # both
    sd[KM] = out["dense_melW"][0][KM]
    out["perturbed_stft_dense_melW"] = (sd, True)
    return out


@pytest.mark.parametrize("name", ["hamming_window", "perturbed_stft", "dense_melW", "perturbed_stft_dense_melW"])
def test_stored_frontend_buffers_are_applied(synth_sd, name):
    sd, want_dense = variants(synth_sd)[name]
    m = make(sd)
    dev = torch.device("cuda", 0)
    ctx = m.native_context(dev)
    info = ctx.frontend_info()
    assert info["dense_dft"] is want_dense, info
    if name.endswith("dense_melW"):
        assert info["mel_taps"] == 513 * 224
    wav = synth.synth_waveforms(3, 48000, seed=77)
    B, L = wav.shape
    T = _ffi.num_frames(L)
    # the log-mel tap through the per-kernel entry point (dense path: the context's own scratch) ...
    out = torch.empty(B, T, 224, device="cuda")
    wd = wav.cuda()
    _ffi.check(_ffi.lib().acx_logmel_bn0(ctx.handle, _ffi.ptr(wd), B, L, _ffi.ptr(out), 0, _ffi.stream_ptr(dev)))
    ref = ref_cpu.logmel(sd, ref_cpu.spectrogram(sd, wav))[:, 0]
    d = (out.cpu() - ref).abs()
    strong = ref > (ref.max() - 90.0)                      # same dB criterion as test_logmel_kernel
    assert float(d[strong].max()) < 0.02, float(d[strong].max())
    assert float(d.mean()) < 0.05
    # ... and the whole forward (caller's workspace), all three outputs, against the oracle on the same weights
    got = m(wd)
    frame = m.forward_frame_embeddings(wd)
    scene = m.forward_scene_embeddings(wd)
    torch.cuda.synchronize()
    want = ref_cpu.forward(sd, wav)
    assert float((got["clipwise_logits"].cpu() - want["clipwise_logits"]).abs().max()) < 1e-3
    assert float((got["clipwise_output"].cpu() - want["clipwise_output"]).abs().max()) < 1e-3
    assert float((frame.cpu() - ref_cpu.forward_frame_embeddings(sd, wav)).abs().max()) < 1e-3
    assert float((scene.cpu() - ref_cpu.forward_scene_embeddings(sd, wav)).abs().max()) < 1e-3


def test_dense_frontend_matches_fft_frontend_on_the_standard_tables(synth_sd):
    """A perturbation below the acceptance threshold keeps the FFT; one just above it switches to the dense contraction: the two
    frontends agree on the log-mel of the same clip (two implementations of one contraction), and a batch larger than the
    sub-batch threshold runs the dense path on the split streams as well."""
    g = torch.Generator().manual_seed(5)
    sd = dict(synth_sd)
    sd[KR] = synth_sd[KR] + 1e-5 * torch.randn(synth_sd[KR].shape, generator=g)       # > 2e-6: dense
    m_fft, m_dense = make(synth_sd), make(sd)
    dev = torch.device("cuda", 0)
    assert not m_fft.native_context(dev).frontend_info()["dense_dft"]
    assert m_dense.native_context(dev).frontend_info()["dense_dft"]
    wav = synth.synth_waveforms(20, 32000, seed=3).cuda()
    a, b = m_fft(wav)["clipwise_logits"], m_dense(wav)["clipwise_logits"]
    torch.cuda.synchronize()
    assert float((a - b).abs().max()) < 1e-3
    one = m_dense(wav[7:8])["clipwise_logits"]
    assert torch.equal(one[0], b[7])                       # clip-independent, also through the GEMM frontend
