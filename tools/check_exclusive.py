#!/usr/bin/env python
"""Build-time check of the CU-exclusive launch contract (acx_internal.h): every kernel that runs dense
16-bit MFMA (gemm_split_kernel, mlp_fused_split_kernel, mlp_fused_wide*_kernel, gemm_bf16_kernel, ...) must ship with a
register allocation that fills the SIMD -- 256 registers per lane for 512-thread workgroups, 512 for 256-thread ones.

Reads the kernel metadata of the code objects INSIDE the built libacx.so (llvm-objdump --offloading + llvm-readelf
--notes): what is checked is what ships, whatever flags or toolchain built it.  No GPU needed.
    python tools/check_exclusive.py [path/to/libacx.so]            exit code 0 = contract holds"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("ACX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
# kernels (namespace acx) that run dense 16-bit MFMA on live data
EXCLUSIVE = ("gemm_split_kernel", "gemm_split16_kernel", "mlp_fused_split_kernel", "mlp_fused_wide_kernel", "mlp_fused_wide_bf16_kernel",
             "mlp_fused_stat_bf16_kernel", "gemm_bf16_kernel")


def kernels_of(lib):
    """[(mangled name, threads, vgpr_count incl. AGPRs, scratch bytes)] of every gfx950 kernel inside lib"""
    out = []
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=d, text=True)
            for entry in re.split(r"\n\s+- \.", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", entry)
                if not name or ".vgpr_count" not in entry:
                    continue
                get = lambda k: int(re.search(r"%s:\s+(\d+)" % re.escape(k), entry).group(1))
                out.append((name.group(1), get("max_flat_workgroup_size"), get(".vgpr_count"), get("private_segment_fixed_size")))
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "audioset-convnext-inf_amd", "libacx.so")
    ks = kernels_of(lib)
    bad = seen = 0
    for name, threads, vgpr, scratch in sorted(ks):
        m = re.match(r"_ZN3acx\d+([A-Za-z0-9_]+?)I", name)          # acx::<kernel><template args>
        if not m or m.group(1) not in EXCLUSIVE:
            continue
        seen += 1
        want = 256 if threads == 512 else 512 if threads == 256 else -1
        ok = vgpr == want          # scratch (a few spilled dwords outside the loops) is reported, not an error
        bad += not ok
        print("%s %-100s threads %4d registers %3d (want %3d) scratch %d" % ("ok " if ok else "BAD", name[:100], threads, vgpr, want, scratch))
    if seen < 10:
        print("BAD only %d CU-exclusive kernels found in %s" % (seen, lib))
        bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
