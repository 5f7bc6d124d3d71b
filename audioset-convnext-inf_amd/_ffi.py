"""ctypes binding of libacx.so (include/acx.h).  Plumbing only: torch tensors are handed over as raw
device pointers + sizes; the library never sees a torch type.

The library is built in-tree (`audioset-convnext-inf_amd/libacx.so`, see `__graft_entry__.build()`).
If it is missing, importing this module still works (CPU-only hosts can inspect the package), but
any attempt to *use* the product path raises -- there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACX_LIB") or os.path.join(_HERE, "libacx.so")     # ACX_LIB: diagnostic builds (tools/)

OK = 0
MODE_LOGITS, MODE_SCENE, MODE_FRAME = 0, 1, 2
KERNEL_CLASSES = ("frontend", "stem", "dwconv", "pw1", "pw2", "rowstats", "downsample", "poolhead", "transpose",
                  "mlp_fused", "mlp_wide")
MIN_SAMPLES = 7360

_c_int, _c_i64, _c_sz, _vp = ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p
_pint = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); mirrors include/acx.h one to one
SIGNATURES = {
    "acx_last_error": (ctypes.c_char_p, []),
    "acx_version": (_c_int, []),
    "acx_create": (_c_int, [_c_int, ctypes.POINTER(_vp)]),
    "acx_destroy": (None, [_vp]),
    "acx_set_weight": (_c_int, [_vp, ctypes.c_char_p, _vp, ctypes.POINTER(_c_i64), _c_int]),
    "acx_finalize": (_c_int, [_vp]),
    "acx_set_precision": (_c_int, [_vp, _c_int]),
    "acx_num_frames": (_c_int, [_c_i64, _pint]),
    "acx_stage_hw": (_c_int, [_c_i64, _c_int, _pint, _pint]),
    "acx_workspace_bytes": (_c_int, [_vp, _c_int, _c_i64, _c_int, ctypes.POINTER(_c_sz)]),
    "acx_sub_batches": (_c_int, [_vp, _c_int, ctypes.POINTER(_c_int)]),
    "acx_forward": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_int, _vp, _vp, _vp, _c_sz, _vp]),
    "acx_logmel_bn0": (_c_int, [_vp, _vp, _c_int, _c_i64, _vp, _c_int, _vp]),
    "acx_stem_ln": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _vp]),
    "acx_dwconv7": (_c_int, [_vp, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "acx_dwconv7_bf16": (_c_int, [_vp, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "acx_block": (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _vp, _c_sz, _vp]),
    "acx_block_scratch_bytes": (_c_int, [_c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_sz)]),
    "acx_downsample": (_c_int, [_vp, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "acx_pool_head": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _vp, _vp, _vp]),
    "acx_nhwc_to_nchw": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp]),
    "acx_comm_unique_id": (_c_int, [_vp]),
    "acx_comm_init": (_c_int, [_vp, _c_int, _c_int, _vp]),
    "acx_allgather": (_c_int, [_vp, _vp, _vp, _c_sz, _vp]),
    "acx_comm_info": (_c_int, [_vp, _pint, _pint]),
    "acx_pcm16_to_f32": (_c_int, [_vp, _vp, _c_i64, _vp]),
    "acx_frontend_info": (_c_int, [_vp, _pint, ctypes.POINTER(ctypes.c_float), _pint]),
    "acx_set_frontend": (_c_int, [_vp, _c_int]),
    "acx_tuning_refresh": (_c_int, []),
    "acx_test_fail_sub": (_c_int, [_vp, _c_int]),
    "acx_profile_enable": (_c_int, [_vp, _c_int]),
    "acx_profile_read": (_c_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_i64)]),
}

_lib = None


class AcxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libacx error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load libacx.so once.  Raises if the HIP library has not been built: no fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "libacx.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C audioset-convnext-inf_amd/csrc). The HIP library IS the product path; "
                "there is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc):
    if rc != OK:
        raise AcxError(rc, lib().acx_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Raw pointer of a torch tensor (must be contiguous) or None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor handed to libacx must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


PRECISIONS = {"fp32": 0, "bf16": 1, "fp32_split": 2, "bf16a": 3}
FRONTENDS = {"auto": 0, "dense": 1}      # enum acx_frontend


class Context:
    """Owns one acx_ctx (repacked weights on one GPU)."""

    def __init__(self, device_index):
        self._h = _vp()
        check(lib().acx_create(int(device_index), ctypes.byref(self._h)))
        self.device_index = int(device_index)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().acx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        # at interpreter shutdown the module globals (lib, _vp) may already be gone: the process is about to
        # release the device anyway
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def handle(self):
        return self._h

    def set_precision(self, precision):
        """"fp32" (default, the parity path) or "bf16" (bf16 MFMA operands, fp32 accumulate / LayerNorm / residual).
        Takes effect at the next load_state_dict()."""
        check(lib().acx_set_precision(self._h, PRECISIONS[precision]))

    def load_state_dict(self, sd):
        """sd: mapping key -> tensor (any device); fp32 tensors are staged through host memory."""
        import torch
        l = lib()
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            t = v.detach().to(device="cpu", dtype=torch.float32).contiguous()
            shape = (ctypes.c_int64 * max(1, t.dim()))(*t.shape)
            check(l.acx_set_weight(self._h, k.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()))
        check(l.acx_finalize(self._h))

    def workspace_bytes(self, B, L, mode):
        out = _c_sz()
        check(lib().acx_workspace_bytes(self._h, int(B), int(L), int(mode), ctypes.byref(out)))
        return out.value

    def sub_batches(self, B):
        """How many sub-batches (on separate streams) a forward of B clips runs as (acx_sub_batches)."""
        out = _c_int()
        check(lib().acx_sub_batches(self._h, int(B), ctypes.byref(out)))
        return out.value

    # ---- the C ABI's own collective (include/acx.h): RCCL all-gather without torch.distributed ----
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(128)
        check(lib().acx_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        check(lib().acx_comm_init(self._h, int(rank), int(world), ctypes.c_char_p(unique_id)))

    def allgather(self, local):
        """(n, ...) contiguous device tensor -> (world * n, ...) in rank order, on the current stream."""
        import torch
        r, w = _c_int(), _c_int()
        check(lib().acx_comm_info(self._h, ctypes.byref(r), ctypes.byref(w)))
        out = torch.empty((w.value * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        check(lib().acx_allgather(self._h, ptr(local), ptr(out), local.numel() * local.element_size(), stream_ptr(local.device)))
        return out

    def set_frontend(self, mode):
        """"auto" (FFT kernel when the stored STFT buffers are window x DFT) or "dense" (always the reference's dense DFT
        contraction with the stored weights: the parity mode, acx.h acx_set_frontend)."""
        check(lib().acx_set_frontend(self._h, FRONTENDS[mode]))

    def frontend_info(self):
        """{"dense_dft": bool, "stft_deviation": float, "mel_taps": int} -- how acx_finalize evaluates the frontend."""
        d, dev, taps = _c_int(), ctypes.c_float(), _c_int()
        check(lib().acx_frontend_info(self._h, ctypes.byref(d), ctypes.byref(dev), ctypes.byref(taps)))
        return {"dense_dft": bool(d.value), "stft_deviation": dev.value, "mel_taps": taps.value}

    def profile(self, on):
        check(lib().acx_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        ms = (ctypes.c_double * len(KERNEL_CLASSES))()
        n = (ctypes.c_int64 * len(KERNEL_CLASSES))()
        check(lib().acx_profile_read(self._h, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(KERNEL_CLASSES)}


def stage_hw(L, stage):
    h, w = _c_int(), _c_int()
    check(lib().acx_stage_hw(int(L), int(stage), ctypes.byref(h), ctypes.byref(w)))
    return h.value, w.value


def num_frames(L):
    t = _c_int()
    check(lib().acx_num_frames(int(L), ctypes.byref(t)))
    return t.value
