"""Restatement of the frontend constants of torchlibrosa 0.0.9 / librosa 0.8.1.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The library source is NOT under
/root/reference (pins: environment.yml:48 librosa==0.8.1, :71 torchlibrosa==0.0.9);
this file restates its published algorithm in numpy, step by step in the order the
library performs it, for the one configuration the reference constructs
(convnext.py:161-200): hann window, n_fft = win = 1024, hop 320, center/reflect,
sr 32 kHz, 224 slaney mels in [50, 14000] Hz, power 2, ref 1.0, amin 1e-10, no top_db.

Parity: unpinned by the reference (no tests / vectors at this boundary).  A real
checkpoint stores these very tables (`spectrogram_extractor.stft.conv_{real,imag}.weight`,
`logmel_extractor.melW`) and is the strong pin once available.
"""
import numpy as np

N_FFT = 1024
HOP = 320
SR = 32000
N_MELS = 224
FMIN = 50.0
FMAX = 14000.0
AMIN = 1e-10
REF = 1.0


def hann_periodic(n=N_FFT):
    """scipy.signal.get_window('hann', n, fftbins=True) == librosa.filters.get_window."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def dft_matrix(n=N_FFT):
    """torchlibrosa DFTBase.dft_matrix: W = omega ** (x*y), omega = exp(-2 pi i / n)."""
    x, y = np.meshgrid(np.arange(n), np.arange(n))
    omega = np.exp(-2 * np.pi * 1j / n)
    return np.power(omega, x * y)


def stft_conv_weights(n=N_FFT):
    """conv_real/conv_imag weights of torchlibrosa.stft.STFT, shape (n//2+1, 1, n) float32.

    weight[k, 0, t] = Re/Im( W[t, k] * window[t] ); pad_center is a no-op (win == n_fft).
    """
    w = hann_periodic(n)
    W = dft_matrix(n)
    out = n // 2 + 1
    prod = W[:, 0:out] * w[:, None]
    real = np.real(prod).T.astype(np.float32)[:, None, :]
    imag = np.imag(prod).T.astype(np.float32)[:, None, :]
    return np.ascontiguousarray(real), np.ascontiguousarray(imag)


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_min, f_sp = 0.0, 200.0 / 3
    mels = (f - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        m = f >= min_log_hz
        mels[m] = min_log_mel + np.log(f[m] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_min, f_sp = 0.0, 200.0 / 3
    freqs = f_min + f_sp * m
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if m.ndim:
        sel = m >= min_log_mel
        freqs[sel] = min_log_hz * np.exp(logstep * (m[sel] - min_log_mel))
    elif m >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (m - min_log_mel))
    return freqs


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (htk=False, norm='slaney',
    dtype=float32), shape (n_mels, 1+n_fft//2)."""
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def melW():
    """LogmelFilterBank.melW = librosa.filters.mel(...).T, shape (513, 224) float32."""
    return np.ascontiguousarray(mel_filterbank().T)
