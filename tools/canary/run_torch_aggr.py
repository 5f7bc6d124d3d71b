"""Is it us?  torch's own GEMMs (hipBLASLt / rocBLAS) as the aggressor on the null stream, rocFFT as the victim."""
import torch
side = torch.cuda.Stream()
v = torch.randn(2048, 1024, device="cuda")
ref = torch.fft.rfft(v, dim=1).abs(); torch.cuda.synchronize()
for dt in (torch.float16, torch.bfloat16, torch.float32):
    a = torch.randn(8192, 8192, device="cuda", dtype=dt); b = torch.randn(8192, 8192, device="cuda", dtype=dt)
    torch.matmul(a, b); torch.cuda.synchronize()
    bad = 0
    for it in range(20):
        torch.cuda.synchronize()
        for _ in range(3): c = torch.matmul(a, b)
        with torch.cuda.stream(side):
            got = torch.fft.rfft(v, dim=1).abs()
        torch.cuda.synchronize()
        bad += int(not torch.equal(got, ref))
    print("torch.matmul %s on the null stream: rocFFT rfft on a side stream wrong in %d of 20 runs" % (dt, bad))
