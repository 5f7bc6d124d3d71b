"""Band-limited sinc resampling as `torchaudio.functional.resample` performs it -- the call the reference's demo
makes when a clip is not at 32 kHz (demo_convnext.py:53-59, default arguments: Hann-windowed sinc,
lowpass_filter_width 6, rolloff 0.99).

torchaudio is a pip dependency of the reference (pinned ==0.11.0, environment.yml:70), absent from /root/reference
and from this image: this is a restatement of its published algorithm (torchaudio 0.11 `_get_sinc_resample_kernel` /
`_apply_sinc_resample_kernel`), built from the same torch CPU operations in the same order and dtype (the kernel is
evaluated in the waveform's dtype, float32), so on a machine that has torchaudio the two agree to float rounding.
Parity is UNPINNED by the reference (it holds no vector for this step); tests/test_next_rows_cpu.py checks the
restatement against a direct float64 evaluation of the interpolation formula and against analytic tones.

Host-side I/O, like the reference (which resamples on the CPU before `.to(device)`): not part of the GPU hot path.
"""
import math

import torch


def sinc_resample_kernel(orig_freq, new_freq, gcd, lowpass_filter_width=6, rolloff=0.99, dtype=torch.float32):
    """(new_freq/gcd, 1, 2*width + orig_freq/gcd) filter bank and `width`: output phase i of every block of
    new_freq/gcd output samples is one windowed-sinc FIR over the input."""
    if not (int(orig_freq) == orig_freq and int(new_freq) == new_freq):
        raise Exception("Frequencies must be of integer type to ensure quality resampling computation.")
    orig_freq = int(orig_freq) // gcd
    new_freq = int(new_freq) // gcd
    assert lowpass_filter_width > 0
    base_freq = min(orig_freq, new_freq)
    base_freq *= rolloff                      # cut-off slightly below Nyquist of the lower rate
    width = math.ceil(lowpass_filter_width * orig_freq / base_freq)
    idx = torch.arange(-width, width + orig_freq, dtype=dtype)
    kernels = []
    for i in range(new_freq):
        t = (-i / new_freq + idx / orig_freq) * base_freq
        t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2       # Hann window over the clamped support
        t *= math.pi
        kernel = torch.where(t == 0, torch.tensor(1.0).to(t), torch.sin(t) / t)
        kernel.mul_(window)
        kernels.append(kernel)
    scale = base_freq / orig_freq
    kernels = torch.stack(kernels).view(new_freq, 1, -1).mul_(scale)
    return kernels, width


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """waveform (..., time) float tensor at `orig_freq` Hz -> (..., ceil(time * new_freq / orig_freq)) at `new_freq`."""
    assert orig_freq > 0.0 and new_freq > 0.0
    if orig_freq == new_freq:
        return waveform
    gcd = math.gcd(int(orig_freq), int(new_freq))
    kernel, width = sinc_resample_kernel(orig_freq, new_freq, gcd, lowpass_filter_width, rolloff, waveform.dtype)
    kernel = kernel.to(waveform.device)
    of, nf = int(orig_freq) // gcd, int(new_freq) // gcd
    shape = waveform.size()
    w = waveform.reshape(-1, shape[-1])
    num_wavs, length = w.shape
    w = torch.nn.functional.pad(w, (width, width + of))
    out = torch.nn.functional.conv1d(w[:, None], kernel, stride=of)
    out = out.transpose(1, 2).reshape(num_wavs, -1)
    target_length = int(math.ceil(nf * length / of))
    out = out[..., :target_length]
    return out.view(shape[:-1] + out.shape[-1:])
