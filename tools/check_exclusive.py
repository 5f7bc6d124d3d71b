#!/usr/bin/env python
"""Build-time check of the CU-exclusive launch contract (acx_internal.h): every kernel that runs dense
16-bit MFMA (gemm_split_kernel, mlp_fused_split_kernel, mlp_fused_wide*_kernel, gemm_bf16_kernel, ...) must ship with a
register allocation that fills the SIMD -- 256 registers per lane for 512-thread workgroups, 512 for 256-thread ones.
Second check: for EVERY kernel of the library, the highest vector / accumulator register any of its
instructions names lies inside the allocation its descriptor declares (no register over-reach into a neighbour's file).

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
             "mlp_fused_stat_bf16_kernel", "gemm_bf16_kernel", "dwconv7_mfma_kernel")


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


def register_reach(lib):
    """{kernel: (highest v index, highest a index, vgpr_count, agpr_count)} from the disassembly of every code object."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=d, text=True)
            meta = {}
            last_agpr = 0                      # (.agpr_count opens a kernel's entry, .name / .vgpr_count close it; the split also cuts at every argument)
            for entry in re.split(r"\n\s+- \.", notes)[1:]:
                ag = re.match(r"agpr_count:\s+(\d+)", entry)
                if ag:
                    last_agpr = int(ag.group(1))
                name = re.search(r"\.name:\s+(\S+)", entry)
                if not name or ".vgpr_count" not in entry:
                    continue
                get = lambda k, e=entry: int(re.search(r"%s:\s+(\d+)" % re.escape(k), e).group(1))
                meta[name.group(1)] = (get(".vgpr_count"), last_agpr)
            dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f], cwd=d, text=True)
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1) if m.group(1) in meta else None
                    if cur:
                        out[cur] = [-1, -1, meta[cur][0], meta[cur][1]]
                    continue
                if cur is None or "//" not in line and not line.startswith("\t"):
                    continue
                text = line.split("//")[0]
                for kind, idx in ((0, r"\bv(\d+)\b"), (0, r"\bv\[\d+:(\d+)\]"), (1, r"\ba(\d+)\b"), (1, r"\ba\[\d+:(\d+)\]")):
                    for g in re.findall(idx, text):
                        out[cur][kind] = max(out[cur][kind], int(g))
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "audioset-convnext-inf_amd", "libacx.so")
    ks = kernels_of(lib)
    bad = seen = 0
    reach = register_reach(lib)
    over = 0
    for name, (vmax, amax, vg, ag) in sorted(reach.items()):
        # unified file: vgpr_count = architected VGPRs (rounded up to the accumulator offset) + agpr_count
        arch = vg - ag
        if vmax >= arch or amax >= max(ag, 0) and amax >= 0:
            print("BAD register over-reach: %s names v%d / a%d with %d architected + %d accumulator registers allocated" % (name[:90], vmax, amax, arch, ag))
            over += 1
    print("register reach: %d kernels audited, %d over-reaching" % (len(reach), over))
    bad += over
    if len(reach) < 40:
        print("BAD only %d kernels found for the register audit" % len(reach))
        bad += 1
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
