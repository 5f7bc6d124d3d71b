import ctypes, sys, torch
lib = ctypes.CDLL("/tmp/libburn.so"); lib.burn_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
side = torch.cuda.Stream(); out = torch.zeros(4096, device="cuda")
v = torch.randn(8192, 4096, device="cuda")
def victims():
    return torch.softmax(v, dim=1), torch.fft.rfft(v[:2048, :1024], dim=1).abs()
ref = victims(); torch.cuda.synchronize()
for kind, name in ((0, "fp16 MFMA burn"), (1, "bf16 MFMA burn"), (-1, "nothing")):
    bad = [0, 0]
    for it in range(20):
        torch.cuda.synchronize()
        if kind >= 0: lib.burn_launch(kind, out.data_ptr(), 256, 20000, None)      # one workgroup per CU, ~10 ms on the null stream
        with torch.cuda.stream(side):
            got = victims()
        torch.cuda.synchronize()
        for k in range(2): bad[k] += int(not torch.equal(got[k], ref[k]))
    print("%s on the null stream: torch softmax / rfft wrong in %s of 20 runs" % (name, bad))
