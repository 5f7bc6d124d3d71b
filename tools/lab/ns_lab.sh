#!/bin/bash
# On the GPU box: builds and runs the variants of tools/lab/ns_lab.hip (stamps, ablations), full-size and 16x the pixels
set -e
FL="-O3 -std=c++17 -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 --offload-arch=gfx950 -w -Iaudioset-convnext-inf_amd/csrc -Itools/lab"
for v in ${VARIANTS:-"" "-DACX_NS_STAMPS" "-DACX_NS_NOGELU=1" "-DACX_NS_NOMFMA=1"}; do
  /opt/rocm/bin/hipcc $FL $v tools/lab/ns_lab.hip -o /tmp/ns_lab 2>&1 | grep -E "error" || true
  echo "== variant [$v]"
  /tmp/ns_lab; /tmp/ns_lab 903168
  if [ -z "$v" ]; then /tmp/ns_lab 56448 ring; /tmp/ns_lab 903168 ring; fi
done
