#!/bin/sh
# A/B two builds of the library on ONE box: sh tools/ab_lib.sh <other.so> <command...>  runs the command with the in-tree
# library, with <other.so> swapped in, and with the in-tree library again (box-to-box differences exceed most effects).
L=comfyui-float_optimized_amd/csrc/libfloat_hip.so
O=$1; shift
cp $L /tmp/_keep.so
echo "== in-tree"; "$@"
cp $O $L; echo "== $O"; "$@"
cp /tmp/_keep.so $L; echo "== in-tree again"; "$@"
