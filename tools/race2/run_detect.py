"""Sensitive detector (tools/race2/detector.hip, packed and scalar builds) next to a list of aggressors.
    python tools/race2/run_detect.py AGGRESSOR [AGGRESSOR ...]
    aggressors: none down1..3 block0..3 dw0..3 dwm0..2 burnN:BLOCKS (N = 1, 4, 6 independent accumulators)
    dwm: the matrix-pipe depthwise kernel of bf16 activations (needs ACX_PRECISION=bf16a)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
V = os.path.join(ROOT, "build", "variants")
dets = {}
for name in ("packed", "scalar", "pkmuladd"):      # pkmuladd (round 4): -ffp-contract=off -- v_pk_mul_f32 / v_pk_add_f32 only, no v_pk_fma_f32
    if not os.path.isfile(os.path.join(V, "libdetector_%s.so" % name)):
        continue
    d = ctypes.CDLL(os.path.join(V, "libdetector_%s.so" % name))
    d.detector_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    d.detector_launch_lds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dets[name] = d
vc = ctypes.CDLL(os.path.join(V, "libvalucls.so"))
vc.burn_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
side = torch.cuda.Stream(); side_sp = ctypes.c_void_p(side.cuda_stream); null_sp = ctypes.c_void_p(0)
seed = torch.randn(65536, device="cuda"); VB = 4096; ROUNDS = 400
scratch = torch.zeros(4096, device="cuda")
DET_LDS = int(os.environ.get("DET_LDS", "0"))       # bytes of (unused) LDS the detector workgroups claim
from audioset_convnext_inf_amd import _ffi, synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
B = 32
m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
DIMS = (96, 192, 384, 768); HS = (252, 126, 63, 31); WS = (56, 28, 14, 7)
def make_load(what):
    if what == "none":
        return lambda: None
    if what.startswith("burn"):
        nacc, blocks = what[4:].split(":")
        return lambda: vc.burn_launch(int(nacc), scratch.data_ptr(), int(blocks), 1500, null_sp)
    zero = what.endswith("z"); what = what.rstrip("z")
    s = int(what[-1])
    if what.startswith("down"):
        x = (torch.zeros if zero else torch.randn)(B, HS[s - 1], WS[s - 1], DIMS[s - 1], device="cuda"); out = torch.empty(B, HS[s], WS[s], DIMS[s], device="cuda"); scr = torch.empty_like(x)
        def f():
            for _ in range(8): lib.acx_downsample(h, s, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scr), B, HS[s - 1], WS[s - 1], null_sp)
        f.keep = (x, out, scr)
        return f
    if what.startswith("block"):
        need = ctypes.c_size_t(); lib.acx_block_scratch_bytes(s, B, HS[s], WS[s], ctypes.byref(need))
        x = torch.randn(B, HS[s], WS[s], DIMS[s], device="cuda"); scr = torch.empty(need.value, dtype=torch.uint8, device="cuda")
        def f():
            for _ in range(4): lib.acx_block(h, s, 0, _ffi.ptr(x), B, HS[s], WS[s], _ffi.ptr(scr), need.value, null_sp)
        f.keep = (x, scr)
        return f
    if what.startswith("dwm"):
        x = torch.randn(B, HS[s], WS[s], DIMS[s], device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
        def f():
            for _ in range(12): _ffi.check(lib.acx_dwconv7_bf16(h, s, 0, _ffi.ptr(x), _ffi.ptr(y), B, HS[s], WS[s], null_sp))
        f.keep = (x, y)
        return f
    if what.startswith("dw"):
        x = torch.randn(B, HS[s], WS[s], DIMS[s], device="cuda"); y = torch.empty_like(x)
        def f():
            for _ in range(12): lib.acx_dwconv7(h, s, 0, _ffi.ptr(x), _ffi.ptr(y), None, B, HS[s], WS[s], null_sp)
        f.keep = (x, y)
        return f
    raise SystemExit("unknown aggressor " + what)
def detect(name, sp):
    o = torch.empty(VB * 256, device="cuda")
    assert dets[name].detector_launch_lds(seed.data_ptr(), o.data_ptr(), VB, ROUNDS, DET_LDS, sp) == 0
    return o
refs = {n: detect(n, null_sp) for n in dets}; torch.cuda.synchronize()
# round 4: a REAL packed-FP32 victim of the library itself -- the column-streaming depthwise kernel (343 v_pk_fma_f32 per row and
# lane, LDS-DMA, 122 KB of LDS per workgroup: it shares a CU with nothing that needs more than 38 KB)
vx = torch.randn(16, HS[0], WS[0], DIMS[0], device="cuda"); vy = torch.empty_like(vx); vref = torch.empty_like(vx)
os.environ["ACX_DW_STREAM"] = "1"; lib.acx_tuning_refresh()
lib.acx_dwconv7(h, 0, 1, _ffi.ptr(vx), _ffi.ptr(vref), None, 16, HS[0], WS[0], null_sp); torch.cuda.synchronize()
for what in sys.argv[1:]:
    load = make_load(what)
    res = []
    for n in dets:
        bad_runs = bad_thr = 0
        for it in range(5):
            torch.cuda.synchronize()
            load()
            with torch.cuda.stream(side):
                o = detect(n, side_sp)
            torch.cuda.synchronize()
            nb = int((o != refs[n]).sum()); bad_thr += nb; bad_runs += int(nb > 0)
        res.append("%s detector wrong in %d/5 runs (%d threads)" % (n, bad_runs, bad_thr))
    bad_runs = bad_el = 0
    for it in range(5):
        torch.cuda.synchronize()
        load()
        with torch.cuda.stream(side):
            lib.acx_dwconv7(h, 0, 1, _ffi.ptr(vx), _ffi.ptr(vy), None, 16, HS[0], WS[0], side_sp)
        torch.cuda.synchronize()
        nb = int((vy != vref).sum()); bad_el += nb; bad_runs += int(nb > 0)
    res.append("dwconv7_col victim wrong in %d/5 runs (%d elements)" % (bad_runs, bad_el))
    print("DETECT det_lds=%d lib=%s prec=%s aggressor=%-12s %s" % (DET_LDS, os.path.basename(os.environ.get("ACX_LIB", "libacx.so")),
                                                        os.environ.get("ACX_PRECISION", "fp32_split"), what, "; ".join(res)), flush=True)
