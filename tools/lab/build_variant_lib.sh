#!/bin/bash
# Builds a variant libacx into build/labs/libacx_<name>.so: the current csrc/ with files overridden from a directory.
#   tools/lab/build_variant_lib.sh <name> <override-dir>     (override-dir holds replacement *.hip / *.h files)
set -e
cd "$(dirname "$0")/../.."
name=$1; over=$2
W=build/variants/$name; rm -rf $W; mkdir -p $W/pkg/csrc $W/include $W/build/acx
cp audioset-convnext-inf_amd/csrc/* $W/pkg/csrc/; cp include/acx.h $W/include/
[ -n "$over" ] && cp $over/* $W/pkg/csrc/
make -C $W/pkg/csrc -j8 2>&1 | grep -E "error|Error" || true
cp $W/pkg/libacx.so build/labs/libacx_$name.so
ls -la build/labs/libacx_$name.so
