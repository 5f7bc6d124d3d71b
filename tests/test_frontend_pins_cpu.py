"""CPU: independent pins of the frontend constants.

torchlibrosa / librosa are absent from /root/reference and from this image, so the oracle's frontend is a
restatement (oracle/torchlibrosa_spec.py).  These tests pin it with what the image DOES have -- scipy's window function
and torch.stft -- and with closed-form properties of the Slaney mel scale, none of which share code with the restatement.
Reference call sites: src/audioset_convnext_inf_amd/pytorch/convnext.py:179-200, 298-299.
"""
import math

import numpy as np
import pytest
import scipy.signal
import torch

from oracle import ref_cpu, torchlibrosa_spec as tls
from audioset_convnext_inf_amd import frontend_tables as ft, synth


def test_window_is_scipys_periodic_hann():
    """librosa.filters.get_window('hann', 1024, fftbins=True) IS scipy.signal.get_window(...) (librosa 0.8.1 filters.py)."""
    w = scipy.signal.get_window("hann", 1024, fftbins=True)
    np.testing.assert_allclose(tls.hann_periodic(), w, rtol=0, atol=1e-15)
    # the product tables carry the same window in their bin-0 row (cos = 1): what acx_finalize extracts as d_hann
    real, _ = ft.stft_weights()
    np.testing.assert_allclose(real[0, 0], w.astype(np.float32), rtol=0, atol=6e-8)


@pytest.mark.parametrize("L", [32000, 7360, 96123])
def test_oracle_spectrogram_is_torch_stft_power(synth_sd, L):
    """center = True, pad_mode = 'reflect', hop 320, n_fft = win = 1024, power 2: frame count, padding and bins against
    torch.stft (its own FFT, its own framing).  The Conv1d-DFT of the reference and an FFT differ by fp32 noise only:
    <= 1e-4 relative on every bin that carries at least 1e-3 of the frame's peak power (weaker bins sit in the fp32 noise of either formulation; the float64 check below has no such floor)."""
    wav = synth.synth_waveforms(2, L, seed=11)
    spec = ref_cpu.spectrogram(synth_sd, wav)[:, 0]                               # (B, T, 513)
    win = torch.from_numpy(scipy.signal.get_window("hann", 1024, fftbins=True)).float()
    st = torch.stft(wav, 1024, 320, 1024, window=win, center=True, pad_mode="reflect", return_complex=True)
    ref = (st.abs() ** 2).transpose(1, 2)                                          # (B, T, 513)
    assert spec.shape == ref.shape == (2, L // 320 + 1, 513)
    strong = ref >= 1e-3 * ref.amax(dim=2, keepdim=True)
    rel = ((spec - ref).abs() / ref.clamp_min(1e-30))[strong]
    assert float(rel.max()) <= 1e-4, float(rel.max())
    # and in float64 (the formulation itself, without fp32 noise): the DFT matrix x window of the restatement vs torch.stft
    r64, i64 = [torch.from_numpy(np.asarray(a, dtype=np.float64)) for a in
                (np.real(tls.dft_matrix()[:, :513] * tls.hann_periodic()[:, None]).T, np.imag(tls.dft_matrix()[:, :513] * tls.hann_periodic()[:, None]).T)]
    x = torch.nn.functional.pad(wav.double()[:, None, :], (512, 512), mode="reflect")
    re = torch.nn.functional.conv1d(x, r64[:, None, :], stride=320)
    im = torch.nn.functional.conv1d(x, i64[:, None, :], stride=320)
    win64 = torch.from_numpy(scipy.signal.get_window("hann", 1024, fftbins=True))
    st64 = torch.stft(wav.double(), 1024, 320, 1024, window=win64, center=True, pad_mode="reflect", return_complex=True)
    assert float((re - st64.real).abs().max()) <= 1e-9 and float((im - st64.imag).abs().max()) <= 1e-9


def test_slaney_scale_closed_forms():
    """Slaney mel scale (librosa hz_to_mel, htk = False): linear 200/3 Hz per mel below 1 kHz, logarithmic above with
    step log(6.4) / 27 -- break point at 1000 Hz = mel 15, continuous there; 6400 Hz = mel 42."""
    assert tls._hz_to_mel(1000.0) == pytest.approx(15.0, abs=1e-12)
    assert tls._hz_to_mel(500.0) == pytest.approx(7.5, abs=1e-12)
    assert tls._hz_to_mel(6400.0) == pytest.approx(42.0, abs=1e-12)
    assert tls._mel_to_hz(15.0) == pytest.approx(1000.0, abs=1e-9)
    assert tls._mel_to_hz(42.0) == pytest.approx(6400.0, abs=1e-9)
    below, above = tls._hz_to_mel(999.999), tls._hz_to_mel(1000.001)
    assert 0 < above - below < 1e-4                                  # continuous across the break
    f = np.array([50.0, 300.0, 1000.0, 4000.0, 14000.0])
    np.testing.assert_allclose(tls._mel_to_hz(tls._hz_to_mel(f)), f, rtol=1e-12)


def _tri_filter_by_hand(i):
    """Filter i of librosa.filters.mel(32000, 1024, 224, 50, 14000) from the closed forms alone: 226 band edges equally
    spaced in mel between mel(50) and mel(14000); triangle between edges i, i+1, i+2 sampled at k * 31.25 Hz; area
    normalisation 2 / (f[i+2] - f[i])."""
    def hz2mel(f):
        return f / (200.0 / 3) if f < 1000.0 else 15.0 + math.log(f / 1000.0) / (math.log(6.4) / 27.0)

    def mel2hz(m):
        return m * (200.0 / 3) if m < 15.0 else 1000.0 * math.exp((math.log(6.4) / 27.0) * (m - 15.0))
    m_lo, m_hi = hz2mel(50.0), hz2mel(14000.0)
    edges = [mel2hz(m_lo + (m_hi - m_lo) * j / 225.0) for j in range(226)]
    f0, f1, f2 = edges[i], edges[i + 1], edges[i + 2]
    out = np.zeros(513)
    for k in range(513):
        fk = k * 31.25
        out[k] = max(0.0, min((fk - f0) / (f1 - f0), (f2 - fk) / (f2 - f1))) * 2.0 / (f2 - f0)
    return out


@pytest.mark.parametrize("i", [0, 60, 223])          # the first filter, one across the 1 kHz break, the last
def test_three_mel_filters_by_hand(i):
    m = tls.melW()                                     # (513, 224) float32
    hand = _tri_filter_by_hand(i)
    np.testing.assert_allclose(m[:, i], hand, rtol=2e-6, atol=1e-9)
    # area normalisation: (sum of weights) x bin width ~ 1 for filters wide enough to be sampled well
    if i == 223:
        assert float(m[:, i].sum() * 31.25) == pytest.approx(1.0, abs=0.02)
    # the product table is the same matrix
    np.testing.assert_array_equal(ft.mel_matrix()[:, i], m[:, i])


def test_filter_60_straddles_the_break():
    edges_mel = np.linspace(tls._hz_to_mel(50.0), tls._hz_to_mel(14000.0), 226)
    lo, hi = tls._mel_to_hz(edges_mel[60]), tls._mel_to_hz(edges_mel[62])
    assert lo < 1000.0 < hi, (lo, hi)
