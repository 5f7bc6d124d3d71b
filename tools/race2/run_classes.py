"""Which VALU instruction classes go wrong, next to which aggressor?
    python tools/race2/run_classes.py down2|block2|burn1|burn4|burn6|none [blocks_of_burn]
Victims: tools/race2/valu_classes.hip (pure-register, one instruction class each) on a side stream."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
what = sys.argv[1]; burn_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
vc = ctypes.CDLL(os.path.join(ROOT, "build", "variants", "libvalucls.so"))
vc.cls_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
vc.burn_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
NAMES = ["v_fma_f32", "v_pk_fma_f32", "v_pk_mul/add_f32", "v_fma_f64", "v_mad_u64_u32/lshl_b64", "v_mul/add_f32", "v_pk_mov_b32+pk", "v_pk_fma_f16", "v_pk_mul/add_f32 SGPR-pair src", "v_pk_fma_f32 op_sel/neg VGPR", "v_fma/mul_f32 SGPR src", "v_pk complex product SGPR+modifiers"]
side = torch.cuda.Stream(); side_sp = ctypes.c_void_p(side.cuda_stream); null_sp = ctypes.c_void_p(0)
seed = torch.randn(65536, device="cuda"); VB = 4096; ROUNDS = 600
scratch = torch.zeros(4096, device="cuda")
if what in ("down2", "block2"):
    from audioset_convnext_inf_amd import _ffi, synth
    from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
    B = 32
    m = convnext_tiny(after_stem_dim=[252, 56]); m.load_state_dict(synth.synth_state_dict(0)); m = m.to("cuda").eval()
    ctx = m.native_context(torch.device("cuda", 0)); lib = _ffi.lib(); h = ctx.handle
    if what == "down2":
        x = torch.randn(B, 126, 28, 192, device="cuda"); out = torch.empty(B, 63, 14, 384, device="cuda"); scr = torch.empty_like(x)
        def load():
            for _ in range(8): lib.acx_downsample(h, 2, _ffi.ptr(x), _ffi.ptr(out), _ffi.ptr(scr), B, 126, 28, null_sp)
    else:
        need = ctypes.c_size_t(); lib.acx_block_scratch_bytes(2, B, 63, 14, ctypes.byref(need))
        x = torch.randn(B, 63, 14, 384, device="cuda"); scr = torch.empty(need.value, dtype=torch.uint8, device="cuda")
        def load():
            for _ in range(6): lib.acx_block(h, 2, 0, _ffi.ptr(x), B, 63, 14, _ffi.ptr(scr), need.value, null_sp)
elif what.startswith("burn"):
    nacc = int(what[4:])
    def load():
        assert vc.burn_launch(nacc, scratch.data_ptr(), burn_blocks, 1500, null_sp) == 0
else:
    def load(): pass
def victim(cls, sp):
    o = torch.empty(VB * 256, dtype=torch.int32, device="cuda")
    assert vc.cls_launch(cls, seed.data_ptr(), o.data_ptr(), VB, ROUNDS, sp) == 0
    return o
refs = [victim(c, null_sp) for c in range(len(NAMES))]; torch.cuda.synchronize()
again = [victim(c, null_sp) for c in range(len(NAMES))]; torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(refs, again)), "victims not deterministic on an idle GPU"
res = []
for c in range(len(NAMES)):
    bad_runs = bad_thr = 0
    for it in range(6):
        torch.cuda.synchronize()
        load()
        with torch.cuda.stream(side):
            o = victim(c, side_sp)
        torch.cuda.synchronize()
        nb = int((o != refs[c]).sum()); bad_thr += nb; bad_runs += int(nb > 0)
    res.append("%s: %d/6 runs, %d threads" % (NAMES[c], bad_runs, bad_thr))
print("CLASSES lib=%s aggressor=%s%s -> %s" % (os.path.basename(os.environ.get("ACX_LIB", "libacx.so")), what,
                                              (" x%d blocks" % burn_blocks) if what.startswith("burn") else "", "; ".join(res)))
