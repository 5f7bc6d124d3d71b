"""Build-time contract of the hand-counted waits (no GPU needed: reads the code objects inside the built libacx.so).

The persistent fused MLP of stage 1 (mlp_fused_wide.hip, PERS) keeps 24 extra vector-memory operations in flight across some of its
segment-end waits: the next tile's rows, the residual rows, a tile's stores.  Its `s_waitcnt vmcnt(kPieces + 24)` are correct only if
the compiler leaves those loads where the source puts them -- BEHIND the segment's last LDS-DMA piece and in front of the wait.  This
test reads the shipped ISA (tools/lab/isa_skeleton.py's view: loads, stores, LDS-DMA, waits, barriers in program order) and
requires exactly that order; a toolchain that schedules differently fails here, not as a rare wrong logit on the GPU."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "audioset-convnext-inf_amd", "libacx.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _skeletons(pattern):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "lab", "isa_skeleton.py"), pattern], text=True)
    kernels, cur = {}, None
    for line in out.splitlines():
        if line.startswith("== "):
            cur = line[3:].strip()
            kernels[cur] = []
        elif cur is not None and line.strip():
            kernels[cur].append(re.sub(r"^\s*\d+\s+", "", line).strip())
    return kernels


def _find(seq, sub, start=0):
    for i in range(start, len(seq) - len(sub) + 1):
        if seq[i:i + len(sub)] == sub:
            return i
    return -1


@pytest.mark.skipif(not (os.path.isfile(LIB) and os.path.isfile(OBJDUMP)), reason="needs the built libacx.so and llvm-objdump")
@pytest.mark.parametrize("lnout", ["Lb0", "Lb1"])
def test_persistent_fused_mlp_keeps_its_loads_behind_the_pieces(lnout):
    ks = _skeletons("mlp_fused_wide_kernelILi192ELi1E%sELi2ELb1E" % lnout)
    assert len(ks) == 1, list(ks)
    sk = next(iter(ks.values()))
    assert not any("scratch" in x for x in sk), "the persistent kernel must not spill"
    assert not any(x.startswith("flat_") for x in sk), "the row prefetch must stay in the global address space"
    pieces, rows, ahead, plain, bar = ("global_load_lds_dwordx4  x6", "global_load_dwordx4  x24", "s_waitcnt vmcnt(30)",
                                       "s_waitcnt vmcnt(6)", "s_barrier")
    # segment 2n - 5: six pieces, THEN the 24 row loads of the next tile, then the wait that leaves pieces + rows in flight
    i = _find(sk, [pieces, rows, ahead, bar])
    assert i >= 0, sk
    # segment 2n - 4: the rows are still allowed in flight; segment 2n - 3: they must have landed (plain count)
    j = _find(sk, [pieces, ahead, bar, pieces, plain, bar], i + 4)
    assert j == i + 4, (i, j, sk[i:i + 12])
    # then the 24 residual loads, in front of the last two segments
    assert sk[j + 6] == rows, sk[j:j + 8]
    # the first segment of a tile allows for the previous tile's 24 stores
    k = _find(sk, [pieces, ahead, bar])
    assert 0 <= k < i, (k, i)
