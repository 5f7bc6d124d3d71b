#!/usr/bin/env python
"""Build-time check of the CU-exclusive launch contract (acx_internal.h): every kernel that runs dense
16-bit MFMA (gemm_split_kernel, mlp_fused_split_kernel, gemm_bf16_kernel) must be emitted with a register allocation
that fills the SIMD -- 256 registers per lane for 512-thread workgroups, 512 for 256-thread ones.
Compiles the sources to device assembly (no GPU needed) and reads the kernel descriptors.
    python tools/check_exclusive.py            exit code 0 = contract holds"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "audioset-convnext-inf_amd", "csrc")
# every __global__ of these files runs dense 16-bit MFMA
SOURCES = ["gemm_split.hip", "mlp_fused_split.hip", "mlp_fused_wide.hip", "mlp_fused_wide_bf16.hip", "gemm_bf16.hip"]


def descriptors(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-w", "-DACX_BUILD",
                               "--cuda-device-only", "-S", path, "-o", out] + (["-fno-slp-vectorize"] if "mlp_fused_wide" in path else []))
        text = open(out).read()
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        get = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        res[m.group(1)] = {"vgpr": get("next_free_vgpr"), "scratch": get("private_segment_fixed_size")}
    threads = {}
    meta = text[text.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in text else ""
    for entry in re.split(r"\n  - ", meta)[1:]:
        n = re.search(r"\.name:\s+(\S+)", entry)
        t = re.search(r"\.max_flat_workgroup_size:\s+(\d+)", entry)
        if n and t:
            threads[n.group(1)] = int(t.group(1))
    return res, threads


def main():
    bad = 0
    for src in SOURCES:
        path = os.path.join(CS, src)
        if not os.path.isfile(path):
            continue
        desc, threads = descriptors(path)
        for name, d in sorted(desc.items()):
            t = threads.get(name, 0)
            want = 256 if t == 512 else 512 if t == 256 else -1
            ok = d["vgpr"] == want          # scratch (a few spilled dwords outside the loops) is reported, not an error
            bad += not ok
            print("%s %-90s threads %4d registers %3d (want %3d) scratch %d" % ("ok " if ok else "BAD", name[:90], t, d["vgpr"], want, d["scratch"]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
