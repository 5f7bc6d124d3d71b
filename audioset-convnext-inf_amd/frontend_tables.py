"""Frontend constants owned by the product: the three frozen buffers the reference model
carries in its state_dict (`spectrogram_extractor.stft.conv_{real,imag}.weight` (513,1,1024),
`logmel_extractor.melW` (513,224)), which torchlibrosa builds at construction
(reference call sites convnext.py:179-200).  A checkpoint overwrites them; the module
default (random-init model, as `convnext_tiny()` gives) must already hold the same values.

Closed forms (float64, cast to float32 once):
  conv_real[k,0,n] =  hann[n] cos(2 pi (n k mod N) / N)
  conv_imag[k,0,n] = -hann[n] sin(2 pi (n k mod N) / N),   hann[n] = 0.5 - 0.5 cos(2 pi n / N)
  melW[:, m]       = Slaney-scale, area-normalised triangle between mel points m, m+1, m+2.
"""
import numpy as np

N_FFT = 1024
N_BINS = N_FFT // 2 + 1
HOP = 320
SAMPLE_RATE = 32000
N_MELS = 224
FMIN = 50.0
FMAX = 14000.0

_F_SP = 200.0 / 3.0
_LOG_HZ = 1000.0
_LOG_MEL = _LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hann():
    n = np.arange(N_FFT, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)


def stft_weights():
    n = np.arange(N_FFT, dtype=np.int64)[None, :]
    k = np.arange(N_BINS, dtype=np.int64)[:, None]
    ang = 2.0 * np.pi * ((n * k) % N_FFT).astype(np.float64) / N_FFT
    w = hann()[None, :]
    real = (np.cos(ang) * w).astype(np.float32)
    imag = (-np.sin(ang) * w).astype(np.float32)
    return real[:, None, :].copy(), imag[:, None, :].copy()


def _mel_of_hz(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= _LOG_HZ, _LOG_MEL + np.log(np.maximum(f, 1e-300) / _LOG_HZ) / _LOGSTEP, f / _F_SP)


def _hz_of_mel(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= _LOG_MEL, _LOG_HZ * np.exp(_LOGSTEP * (m - _LOG_MEL)), _F_SP * m)


def mel_matrix():
    """(513, 224) float32; column m is mel filter m."""
    pts = _hz_of_mel(np.linspace(_mel_of_hz(FMIN), _mel_of_hz(FMAX), N_MELS + 2))
    freqs = np.linspace(0.0, SAMPLE_RATE / 2.0, N_BINS)
    lo, mid, hi = pts[:-2, None], pts[1:-1, None], pts[2:, None]
    up = (freqs[None, :] - lo) / (mid - lo)
    down = (hi - freqs[None, :]) / (hi - mid)
    tri = np.maximum(0.0, np.minimum(up, down)).astype(np.float32)      # librosa fills a float32 array
    tri *= (2.0 / (hi - lo))                                             # ... and scales it in place
    return np.ascontiguousarray(tri.T)
