#!/bin/bash
# On the GPU box: ablations of the N-split kernel's skeleton (wrong results, timing only)
FL="-O3 -std=c++17 -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 --offload-arch=gfx950 -w -Iaudioset-convnext-inf_amd/csrc -Itools/lab"
run() { /opt/rocm/bin/hipcc $FL $1 tools/lab/ns_lab.hip -o /tmp/ns_lab && echo "== [$1]" && /tmp/ns_lab && /tmp/ns_lab 903168; }
run ""
run "-DACX_NS_NODMA"
run "-DACX_NS_NOWAIT"
run "-DACX_NS_NOBAR"
run "-DACX_NS_NOREAD"
run "-DACX_NS_NODMA -DACX_NS_NOBAR"
run "-DACX_NS_NODMA -DACX_NS_NOBAR -DACX_NS_NOREAD"
run "-DACX_NS_NODMA -DACX_NS_NOBAR -DACX_NS_NOREAD -DACX_NS_NOGELU=1"
run "-DACX_NS_NODMA -DACX_NS_NOBAR -DACX_NS_NOREAD -DACX_NS_NOGELU=1 -DACX_NS_NOBIAS"
run "-DACX_NS_NOBIAS -DACX_NS_NOGELU=1"
