"""CPU restatement (torch-CPU, fp32) of the reference's inference path.

TEST INFRASTRUCTURE (see oracle/__init__.py): the checker for the HIP path and the
timed `cpu_baseline` of bench.py.  Never imported by the product package.

It is a *functional* restatement over a plain ``state_dict`` (the reference's 190
keys), op for op in the reference's order and layouts (NCHW convs, permutes,
``F.layer_norm``, erf-GELU ...), with no channels_last / compile tricks, so that
its timing stands for "the reference's CPU path".  Every function cites the
reference lines it follows (paths relative to /root/reference).

Pin: `tests/golden/make_goldens.py` runs this next to the reference's own
``ConvNeXt`` class (imported in the build container) on the same weights/inputs and
records the max deviation in ``tests/golden/MANIFEST.json``; `tests/test_oracle.py`
re-checks it against the committed golden vectors.
"""
import torch
import torch.nn.functional as F

DEPTHS = (3, 3, 9, 3)           # convnext.py:655
DIMS = (96, 192, 384, 768)      # convnext.py:656
N_FFT, HOP = 1024, 320          # convnext.py:169-170
MIN_SAMPLES = 7360              # shortest L for which the last 2x2 downsample has input


def spectrogram(sd, wav):
    """torchlibrosa Spectrogram as constructed at convnext.py:179-187, called :298.
    wav (B, L) -> power spectrogram (B, 1, T, 513), T = L // 320 + 1."""
    x = wav[:, None, :]
    x = F.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    real = F.conv1d(x, sd["spectrogram_extractor.stft.conv_real.weight"], stride=HOP)
    imag = F.conv1d(x, sd["spectrogram_extractor.stft.conv_imag.weight"], stride=HOP)
    real = real[:, None, :, :].transpose(2, 3)
    imag = imag[:, None, :, :].transpose(2, 3)
    return real ** 2 + imag ** 2


def logmel(sd, spec):
    """torchlibrosa LogmelFilterBank as constructed at convnext.py:190-200, called :299.
    (B,1,T,513) -> (B,1,T,224) in dB (ref 1.0, amin 1e-10, top_db None)."""
    mel = torch.matmul(spec, sd["logmel_extractor.melW"])
    out = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    out = out - 10.0 * float(torch.log10(torch.tensor(max(1e-10, 1.0))))
    return out


def bn0(sd, x):
    """convnext.py:304-306 -- eval-mode BatchNorm2d(224) over the mel axis."""
    x = x.transpose(1, 3)
    x = F.batch_norm(x, sd["bn0.running_mean"], sd["bn0.running_var"],
                     sd["bn0.weight"], sd["bn0.bias"], training=False, eps=1e-5)
    return x.transpose(1, 3)


def ln_channels_first(x, w, b, eps=1e-6):
    """convnext.py:536-541 (LayerNorm data_format='channels_first')."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def stem(sd, x):
    """downsample_layers[0]: Conv2d(1,96,k4,s4,padding=(4,0)) (convnext.py:688-691,707)
    + LayerNorm channels_first (:227).  (B,1,T,224) -> (B,96,H0,56)."""
    x = F.conv2d(x, sd["downsample_layers.0.0.weight"], sd["downsample_layers.0.0.bias"],
                 stride=(4, 4), padding=(4, 0))
    return ln_channels_first(x, sd["downsample_layers.0.1.weight"], sd["downsample_layers.0.1.bias"])


def downsample(sd, i, x):
    """downsample_layers[i], i=1..3: LN channels_first + Conv2d(k2,s2) (convnext.py:230-235)."""
    p = "downsample_layers.%d." % i
    x = ln_channels_first(x, sd[p + "0.weight"], sd[p + "0.bias"])
    return F.conv2d(x, sd[p + "1.weight"], sd[p + "1.bias"], stride=2)


def block_dwconv(sd, s, j, x):
    """Block.dwconv (convnext.py:58-60, :76)."""
    p = "stages.%d.%d." % (s, j)
    C = x.shape[1]
    return F.conv2d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=C)


def block(sd, s, j, x, taps=None):
    """Block.forward (convnext.py:74-87), drop_path = Identity (:72)."""
    p = "stages.%d.%d." % (s, j)
    C = x.shape[1]
    inp = x
    x = block_dwconv(sd, s, j, x)
    if taps is not None:
        taps["s%d.b%d.dwconv" % (s, j)] = x
    x = x.permute(0, 2, 3, 1)
    x = F.layer_norm(x, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    if taps is not None:
        taps["s%d.b%d.ln" % (s, j)] = x
    x = F.linear(x, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])
    x = F.gelu(x)
    x = F.linear(x, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
    x = sd[p + "gamma"] * x
    x = x.permute(0, 3, 1, 2)
    return inp + x


def forward_features(sd, x, return_frame_embeddings=False, taps=None):
    """ConvNeXt.forward_features (convnext.py:269-285). x = bn0 output (B,1,T,224)."""
    for i in range(4):
        x = stem(sd, x) if i == 0 else downsample(sd, i, x)
        if taps is not None:
            taps["ds%d" % i] = x
        for j in range(DEPTHS[i]):
            x = block(sd, i, j, x, taps if (taps is not None and j == 0) else None)
            if taps is not None and j == 0:
                taps["s%d.b0.out" % i] = x
        if taps is not None:
            taps["stage%d" % i] = x
    if return_frame_embeddings:
        return x
    x = torch.mean(x, dim=3)
    x1, _ = torch.max(x, dim=2)
    x2 = torch.mean(x, dim=2)
    x = x1 + x2
    if taps is not None:
        taps["pooled"] = x
    return F.layer_norm(x, (DIMS[-1],), sd["norm.weight"], sd["norm.bias"], 1e-6)


def frontend(sd, wav, taps=None):
    """convnext.py:298-306: spectrogram -> logmel -> bn0. (B,L) -> (B,1,T,224)."""
    x = spectrogram(sd, wav)
    x = logmel(sd, x)
    if taps is not None:
        taps["logmel"] = x
    x = bn0(sd, x)
    if taps is not None:
        taps["bn0"] = x
    return x


@torch.no_grad()
def forward(sd, wav, taps=None):
    """ConvNeXt.forward in eval mode (convnext.py:287-331)."""
    x = frontend(sd, wav, taps)
    x = forward_features(sd, x, taps=taps)
    if taps is not None:
        taps["scene"] = x
    logits = F.linear(x, sd["head_audioset.weight"], sd["head_audioset.bias"])
    return {"clipwise_output": torch.sigmoid(logits), "clipwise_logits": logits}


@torch.no_grad()
def forward_scene_embeddings(sd, wav):
    """ConvNeXt.forward_scene_embeddings (convnext.py:333-366)."""
    return forward_features(sd, frontend(sd, wav))


@torch.no_grad()
def forward_frame_embeddings(sd, wav):
    """ConvNeXt.forward_frame_embeddings (convnext.py:369-402) -> NCHW (B,768,H3,7)."""
    return forward_features(sd, frontend(sd, wav), return_frame_embeddings=True)


def out_hw(L):
    """Spatial sizes per stage for a clip of L samples: [(H0,56),(H1,28),(H2,14),(H3,7)]."""
    T = L // HOP + 1
    h = (T + 8 - 4) // 4 + 1
    res = [(h, 56)]
    w = 56
    for _ in range(3):
        h, w = h // 2, w // 2
        res.append((h, w))
    return res
