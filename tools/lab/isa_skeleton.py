#!/usr/bin/env python
"""Memory-operation skeleton of one kernel of libacx.so: loads / stores / LDS-DMA / counted waits / barriers / branches in
program order, runs collapsed.  What the hand-counted s_waitcnt vmcnt(N) of the ring kernels must agree with.
    python tools/lab/isa_skeleton.py <kernel-name substring> [more substrings ...]"""
import os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
PAT = re.compile(r"global_load|global_store|flat_load|flat_store|buffer_load|buffer_store|s_waitcnt vmcnt|s_barrier|s_cbranch|scratch_|s_endpgm")
def main():
    want = sys.argv[1:]
    lib = os.path.join(ROOT, "audioset-convnext-inf_amd", "libacx.so")
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", f], cwd=d, text=True)
            cur, rows = None, []
            def flush():
                if cur and all(w in cur for w in want):
                    print("==", cur)
                    prev, cnt, first = None, 0, 0
                    for ln, key in rows + [(0, None)]:
                        if key == prev:
                            cnt += 1
                            continue
                        if prev is not None:
                            print("%6d  %s%s" % (first, prev, "  x%d" % cnt if cnt > 1 else ""))
                        prev, cnt, first = key, 1, ln
            n = 0
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    flush()
                    cur, rows, n = m.group(1), [], 0
                    continue
                n += 1
                if PAT.search(line):
                    t = line.split("//")[0].split()
                    key = t[0]
                    if key.startswith("s_waitcnt") or key.startswith("s_cbranch"):
                        key = " ".join(t[:3]) if key.startswith("s_waitcnt") else t[0]
                    rows.append((n, key))
            flush()
main()
