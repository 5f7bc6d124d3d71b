"""Host-side mirror of the reference model surface for the audio-tagging hot path.

Same names, arguments, dict keys, output shapes and state_dict keys as the reference's
`audioset_convnext_inf.pytorch.convnext` (ConvNeXt :130-511, Block :44-87, LayerNorm :514-541,
convnext_tiny :641-708), so demo_convnext.py / evaluate_convnext_on_audioset.py /
pytorch_utils.forward can switch imports and keep working.  Nothing here computes: the modules
below are parameter containers (they give `state_dict()` / `load_state_dict()` /
`safetensors.torch.load_model` the reference's 190 keys) and the three forwards hand raw device
pointers to libacx (HIP kernels, include/acx.h) on the current torch stream.

Inference only, GPU only: there is no CPU fallback and no training branch (the reference's
augmentations / mixup / DropPath only run under `self.training`, convnext.py:288-313).
"""
import os

import torch
import torch.nn as nn

from .. import _ffi
from .. import frontend_tables as ft

HF_PYTORCH_WEIGHTS_NAME = "model.safetensors"     # convnext.py:29
HF_CONFIG_NAME = "config.yaml"                    # convnext.py:31

_TINY_DEPTHS = [3, 3, 9, 3]
_TINY_DIMS = [96, 192, 384, 768]


class LayerNorm(nn.Module):
    """Parameter container for the reference LayerNorm (both data formats; convnext.py:514-541)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        self.normalized_shape = (normalized_shape,)


class Block(nn.Module):
    """Parameter container for one ConvNeXt block (convnext.py:44-87); computed by K3 + K4."""

    def __init__(self, dim, drop_path=0.0, layer_scale_init_value=1e-6):
        super().__init__()
        if drop_path != 0.0:
            raise NotImplementedError("stochastic depth is a training feature; the inference path uses rate 0")
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)), requires_grad=True)
        self.drop_path = nn.Identity()


class _STFT(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv_real = nn.Conv1d(1, ft.N_BINS, ft.N_FFT, stride=ft.HOP, bias=False)
        self.conv_imag = nn.Conv1d(1, ft.N_BINS, ft.N_FFT, stride=ft.HOP, bias=False)
        real, imag = ft.stft_weights()
        self.conv_real.weight.data = torch.from_numpy(real)
        self.conv_imag.weight.data = torch.from_numpy(imag)
        for p in self.parameters():
            p.requires_grad = False


class Spectrogram(nn.Module):
    """Holds `stft.conv_real.weight` / `stft.conv_imag.weight` like torchlibrosa's module
    (constructed at convnext.py:179-187); evaluated by the FFT inside kernel K1."""

    def __init__(self):
        super().__init__()
        self.stft = _STFT()


class LogmelFilterBank(nn.Module):
    """Holds `melW` (513,224) like torchlibrosa's module (convnext.py:190-200)."""

    def __init__(self):
        super().__init__()
        self.melW = nn.Parameter(torch.from_numpy(ft.mel_matrix()), requires_grad=False)


class ConvNeXt(nn.Module):
    """Drop-in for the reference `ConvNeXt` restricted to what its shipped entry points build:
    ConvNeXt-Tiny, 527 classes, the [252,56] audio stem, 32 kHz / 1024 / 320 / 224-mel frontend."""

    def __init__(self, in_chans=3, num_classes=1000, depths=[3, 3, 9, 3], dims=[96, 192, 384, 768],
                 drop_path_rate=0.0, use_pydub_augment=False, use_roll_augment=False, use_speed_perturb=False,
                 use_torchaudio=False, layer_scale_init_value=1e-6, head_init_scale=1.0):
        super().__init__()
        if list(depths) != _TINY_DEPTHS or list(dims) != _TINY_DIMS or num_classes != 527:
            raise NotImplementedError("the MI355X path implements ConvNeXt-Tiny / 527 classes (convnext_tiny()); "
                                      "got depths=%r dims=%r num_classes=%r" % (depths, dims, num_classes))
        if use_torchaudio:
            raise NotImplementedError("use_torchaudio=True (Kaldi fbank input) is outside the inference contract")
        if layer_scale_init_value <= 0:
            raise NotImplementedError("layer scale (gamma) is part of the state_dict contract")
        self.use_torchaudio = False
        self.use_pydub_augment = use_pydub_augment      # training-only switches, kept for signature parity
        self.use_roll_augment = use_roll_augment
        self.use_speed_perturb = use_speed_perturb

        self.spectrogram_extractor = Spectrogram()
        self.logmel_extractor = LogmelFilterBank()
        self.bn0 = nn.BatchNorm2d(224)
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(
            nn.Conv2d(1, dims[0], kernel_size=(4, 4), stride=(4, 4), padding=(4, 0)),      # the [252,56] stem
            LayerNorm(dims[0], eps=1e-6, data_format="channels_first")))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(
                LayerNorm(dims[i], eps=1e-6, data_format="channels_first"),
                nn.Conv2d(dims[i], dims[i + 1], kernel_size=2, stride=2)))
        self.stages = nn.ModuleList()
        for i in range(4):
            self.stages.append(nn.Sequential(*[
                Block(dim=dims[i], drop_path=0.0, layer_scale_init_value=layer_scale_init_value)
                for _ in range(depths[i])]))
        self.norm = nn.LayerNorm(dims[-1], eps=1e-6)
        self.head_audioset = nn.Linear(dims[-1], num_classes)
        self.apply(self._init_weights)
        self.head_audioset.weight.data.mul_(head_init_scale)
        self.head_audioset.bias.data.mul_(head_init_scale)

        self._ctx = {}          # device index -> (_ffi.Context, weight signature)
        # arithmetic of the dense contractions (include/acx.h, enum acx_precision).  "fp32_split" and "fp32" are both
        # fp32-grade (same parity tests, same tolerances); split is the fast one.  ACX_PRECISION overrides the default.
        self.precision = os.environ.get("ACX_PRECISION", "fp32_split")
        self.frontend = os.environ.get("ACX_FRONTEND", "auto")       # set_frontend()
        if self.precision not in _ffi.PRECISIONS:
            raise ValueError("ACX_PRECISION must be one of %s" % sorted(_ffi.PRECISIONS))
        self._ws = {}           # (device index, stream handle) -> workspace tensor: concurrent forwards on different
        #                         streams never share scratch memory (the reference module is re-entrant in eval).
        #                         Insertion-ordered, most recently used last; at most _WS_MAX_STREAMS entries are kept
        self._ws_captured = set()   # data pointers of workspaces a stream capture has seen: a hipGraph may point into them
        self._ws_retired = []   # replaced workspaces that a captured graph may still reference (only those) stay alive
        self._weights_epoch = 0  # bumped by load_state_dict / _apply / refresh(): forces a repack of the native weights
        self._sig_cache = None  # (epoch, [parameter and buffer tensors]) -- see _signature

    def _init_weights(self, m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------------------ native side
    def _signature(self):
        """Changes whenever the weights may have: the epoch counter (load_state_dict / _apply / refresh) plus, for every
        parameter and buffer, its identity, storage address and autograd version, plus the identity of every submodule.
        What is cached per epoch is the list of the modules' own `_parameters` / `_buffers` / `_modules` dicts, not the
        tensors: a replaced Parameter (`m.head_audioset.weight = nn.Parameter(...)`) shows up as a new id in its module's
        dict, a replaced submodule (`m.head_audioset = nn.Linear(...)`) as a new child id in its parent's -- the latter
        also rebuilds the cached list.  state_dict() itself builds 190 prefixed keys and cost 0.6 ms per
        forward, ten times the rest of the host path at batch 1 (profiles/r03_f_latency_bs1.txt); this walk costs ~50 us."""
        for _ in range(2):
            c = self._sig_cache
            if c is None or c[0] != self._weights_epoch:
                dicts = [(m._parameters, m._buffers, m._modules) for m in self.modules()]
                c = self._sig_cache = (self._weights_epoch, dicts,
                                       tuple(id(x) for _, _, ch in dicts for x in ch.values()))
            sig = [self.precision, self.frontend, self._weights_epoch]
            kids = []
            for params, bufs, children in c[1]:
                for t in params.values():
                    if t is not None:
                        sig += (id(t), t.data_ptr(), t._version)
                for t in bufs.values():
                    if t is not None:
                        sig += (id(t), t.data_ptr(), t._version)
                kids += [id(x) for x in children.values()]
            if tuple(kids) == c[2]:
                return tuple(sig)
            self._weights_epoch += 1        # the module tree changed: walk it again
        raise RuntimeError("module tree changed while its signature was taken")

    def refresh(self):
        """Call after editing weights in a way autograd's version counters cannot see (`param.data.mul_(...)`, writes
        through numpy views): the native contexts repack the weights at the next forward.  `load_state_dict`, `.to()`,
        `.half()`-style `_apply` calls and ordinary in-place tensor ops are detected without it."""
        self._weights_epoch += 1
        return self

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._weights_epoch += 1
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._weights_epoch += 1
        # contexts and workspaces of devices the module has left are released
        dev = self.head_audioset.weight.device
        keep = dev.index if dev.type == "cuda" and dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else None)
        for idx in [i for i in self._ctx if i != keep]:
            self._ctx.pop(idx)[0].close()
        for key in [k for k in self._ws if k[0] != keep]:
            self._ws.pop(key)
        if keep is None:
            self._ws_retired.clear()
            self._ws_captured.clear()
        return out

    def __getstate__(self):
        # native handles (ctypes) and scratch tensors are per-process state, rebuilt on demand
        state = self.__dict__.copy()
        state["_ctx"], state["_ws"], state["_ws_retired"], state["_ws_captured"], state["_sig_cache"] = {}, {}, [], set(), None
        return state

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_ctx", "_ws"):
                new.__dict__[k] = {}
            elif k == "_ws_retired":
                new.__dict__[k] = []
            elif k == "_ws_captured":
                new.__dict__[k] = set()
            elif k == "_sig_cache":
                new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def set_precision(self, precision):
        """"fp32_split" (default): fp32 GEMM operands carried as fp16 hi + fp16 lo, three fp16 MFMAs per product, fp32
        accumulate -- fp32-grade (1e-3 parity), 16/3 of the f32-MFMA rate.  "fp32": v_mfma_f32_32x32x2_f32 on the
        fp32 operands themselves.  "bf16": pointwise / downsample contractions with bf16 operands and fp32
        accumulation (NOT within the 1e-3 bar); LayerNorm, residual stream, depthwise conv, frontend and head stay
        fp32 in every mode.  "bf16a": "bf16" plus the activations of stages 0-2 (residual stream, depthwise-conv output) stored
        in HBM as bf16 -- half the activation bytes; statistics, accumulation, GELU and the residual add stay fp32.
        The reference has no such switch (closest: torch autocast around its nn.Linear layers)."""
        if precision not in _ffi.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_ffi.PRECISIONS))
        self.precision = precision
        return self

    def set_frontend(self, mode):
        """"auto" (default): the FFT kernel evaluates torchlibrosa's STFT whenever the stored `conv_real` / `conv_imag` buffers
        are window x DFT.  "dense": always the reference's own formulation -- the two Conv1d as one dense contraction with the
        stored weights (convnext.py:179-187,298) -- and thereby its rounding: the parity mode for pure tones and clean sweeps,
        whose bins 90 dB under the frame peak the two formulations round differently (include/acx.h, acx_set_frontend)."""
        if mode not in _ffi.FRONTENDS:
            raise ValueError("frontend must be one of %s" % sorted(_ffi.FRONTENDS))
        self.frontend = mode
        return self

    def native_context(self, device):
        """The libacx context holding this module's weights on `device` (rebuilt when they change)."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        sig = self._signature()
        hit = self._ctx.get(idx)
        if hit is not None and hit[1] == sig:
            return hit[0]
        ctx = hit[0] if hit is not None else _ffi.Context(idx)
        ctx.set_precision(self.precision)
        ctx.set_frontend(self.frontend)
        ctx.load_state_dict(self.state_dict())
        self._ctx[idx] = (ctx, sig)
        return ctx

    _WS_MAX_STREAMS = 8      # workspaces kept per module (one per (device, stream) in use); the least recently used goes first

    def _workspace(self, device, nbytes):
        """Scratch memory of one forward on (device, current stream).  Grows geometrically (a length-sorted sweep would
        otherwise reallocate at every new maximum); a replaced or evicted workspace is freed unless a stream capture has
        seen it -- a captured hipGraph replays with the pointers it recorded, so those stay alive with the module."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.pop(key, None)
        if ws is None or ws.numel() < nbytes:
            if ws is not None:
                self._retire(ws)
                nbytes = max(nbytes, ws.numel() + ws.numel() // 2)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._ws[key] = ws                      # (re-)inserted last: most recently used
        while len(self._ws) > self._WS_MAX_STREAMS:
            self._retire(self._ws.pop(next(iter(self._ws))))
        if torch.cuda.is_current_stream_capturing():
            self._ws_captured.add(ws.data_ptr())
        return ws

    def _retire(self, ws):
        if ws.data_ptr() in self._ws_captured:
            self._ws_retired.append(ws)

    def _run(self, x, mode):
        if self.training:
            raise RuntimeError("inference-only path: call model.eval() first (the reference's training branches -- "
                               "augmentations, SpecAugment, mixup -- are not part of this build)")
        if not isinstance(x, torch.Tensor) or x.dim() != 2:
            raise ValueError("expected a (batch, samples) waveform tensor, got %r" % (getattr(x, "shape", type(x)),))
        wdev = self.head_audioset.weight.device
        if wdev.type != "cuda" or x.device.type != "cuda":
            raise RuntimeError("the MI355X path runs on the GPU only (model on %s, input on %s): "
                               "move both with .to('cuda'); there is no CPU fallback" % (wdev, x.device))
        if x.device != wdev:
            raise RuntimeError("input on %s but model on %s" % (x.device, wdev))
        x = x.detach().to(torch.float32).contiguous()
        B, L = x.shape
        if L < _ffi.MIN_SAMPLES:
            raise RuntimeError("clip of %d samples is too short: kernel size can't be greater than actual input size "
                               "(minimum is %d samples)" % (L, _ffi.MIN_SAMPLES))
        with torch.cuda.device(x.device):
            ctx = self.native_context(x.device)
            ws = self._workspace(x.device, ctx.workspace_bytes(B, L, mode))
            if mode == _ffi.MODE_LOGITS:
                out0 = torch.empty((B, 527), dtype=torch.float32, device=x.device)
                out1 = torch.empty((B, 527), dtype=torch.float32, device=x.device)
            elif mode == _ffi.MODE_SCENE:
                out0, out1 = torch.empty((B, 768), dtype=torch.float32, device=x.device), None
            else:
                h3, w3 = _ffi.stage_hw(L, 3)
                out0, out1 = torch.empty((B, 768, h3, w3), dtype=torch.float32, device=x.device), None
            _ffi.check(_ffi.lib().acx_forward(ctx.handle, _ffi.ptr(x), B, L, mode, _ffi.ptr(out0), _ffi.ptr(out1),
                                              _ffi.ptr(ws), ws.numel(), _ffi.stream_ptr(x.device)))
        return out0, out1

    # ----------------------------------------------------------------------------- public surface
    def forward(self, x, mixup_lambda=None):
        """(B, L) waveform -> {"clipwise_output": probs, "clipwise_logits": logits} (convnext.py:287-331)."""
        logits, probs = self._run(x, _ffi.MODE_LOGITS)
        return {"clipwise_output": probs, "clipwise_logits": logits}

    def forward_scene_embeddings(self, x, mixup_lambda=None):
        """(B, L) -> (B, 768) (convnext.py:333-366)."""
        return self._run(x, _ffi.MODE_SCENE)[0]

    def forward_frame_embeddings(self, x, mixup_lambda=None):
        """(B, L) -> NCHW (B, 768, T', 7) (convnext.py:369-402)."""
        return self._run(x, _ffi.MODE_FRAME)[0]

    @classmethod
    def from_pretrained(cls, pretrained_checkpoint_path, map_location=None, use_auth_token=None):
        """Local file first, then a Zenodo URL, then a Hugging Face model id[@revision]
        (convnext.py:404-511).  Accepts both `model.safetensors` and the `.pth` ({"model": sd}) form that
        evaluate_convnext_on_audioset.py:36-38 loads.  Returns None when the HF repo does not exist."""
        if os.path.isfile(pretrained_checkpoint_path):
            print("Ckpt already on local disk")
            path_ = pretrained_checkpoint_path
        elif "https" in pretrained_checkpoint_path:
            print("Using ckpt from Zenodo")
            dpath_ = os.path.join(torch.hub.get_dir(), "checkpoints")
            os.makedirs(dpath_, exist_ok=True)
            fname = os.path.basename(pretrained_checkpoint_path).replace("?download=1", "")
            path_ = os.path.join(dpath_, fname)
            torch.hub.download_url_to_file(pretrained_checkpoint_path, path_)
        else:
            print("Using ckpt from HF")
            from huggingface_hub import hf_hub_download
            from huggingface_hub.utils import RepositoryNotFoundError
            model_id, _, revision = pretrained_checkpoint_path.partition("@")
            try:
                path_ = hf_hub_download(model_id, HF_PYTORCH_WEIGHTS_NAME, repo_type="model",
                                        revision=revision or None, library_name="audioset-convnext",
                                        token=use_auth_token)
            except RepositoryNotFoundError:
                print("\nCould not download '%s' model.\nIt might be because the model is private or gated so make\n"
                      "sure to authenticate. Visit https://hf.co/settings/tokens to\ncreate your access token and "
                      "retry with use_auth_token=YOUR_AUTH_TOKEN" % model_id)
                return None
            try:        # the hub's download counter keys on config.yaml (convnext.py:470-493)
                hf_hub_download(model_id, HF_CONFIG_NAME, repo_type="model", revision=revision or None,
                                library_name="audioset-convnext", token=use_auth_token)
            except Exception:
                pass
        model = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56],
                              use_speed_perturb=False)
        load_checkpoint(model, path_, map_location or "cpu")
        return model


def load_checkpoint(model, path, map_location="cpu"):
    """`.safetensors` via safetensors.torch.load_model (strict, convnext.py:507); anything else as a torch
    checkpoint holding {"model": state_dict} (evaluate_convnext_on_audioset.py:36-38)."""
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_model as st_load_model
        st_load_model(model, path)
    else:
        ckpt = torch.load(path, map_location=map_location)
        model.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt)
    return model


def convnext_tiny(pretrained=False, strict=False, in_22k=False, drop_path_rate=0.1, after_stem_dim=[56],
                  use_speed_perturb=False, use_pydub_augment=False, use_roll_augment=False, **kwargs):
    """Same signature as the reference factory (convnext.py:641-708).  Only the configuration the
    reference's demo / evaluation / from_pretrained use is built: after_stem_dim=[252, 56], no ImageNet
    pre-training download.  `drop_path_rate` only matters in training and is ignored in eval."""
    if pretrained:
        raise NotImplementedError("ImageNet initialisation is a training feature (and needs network access)")
    if list(after_stem_dim) != [252, 56]:
        raise ValueError("ERROR: this build implements the after_stem_dim=[252,56] stem "
                         "(the one every shipped entry point of the reference uses)")
    return ConvNeXt(in_chans=1, num_classes=527, depths=[3, 3, 9, 3], dims=[96, 192, 384, 768],
                    drop_path_rate=0.0, use_speed_perturb=use_speed_perturb,
                    use_pydub_augment=use_pydub_augment, use_roll_augment=use_roll_augment, **kwargs)
