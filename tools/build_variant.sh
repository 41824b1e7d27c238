#!/bin/sh
# A/B library with extra -D flags for ONE translation unit: sh tools/build_variant.sh <name> <unit.hip> <flags...>
# -> build_ab/<name>.so (the other objects are the in-tree ones; run `make -C comfyui-float_optimized_amd/csrc` first).
# Use with tools/ab_lib.sh build_ab/<name>.so <command...> on the GPU box (same-box alternation).
set -e
NAME=$1; UNIT=$2; shift 2
C=comfyui-float_optimized_amd/csrc
mkdir -p build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed "$@" -c $C/$UNIT -o build_ab/$NAME.o
OBJS=""
for o in misc fmt_api dec_api enc_api aud_api; do
  if [ "$o.hip" = "$UNIT" ]; then OBJS="$OBJS build_ab/$NAME.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/$NAME.so $OBJS
rm -f build_ab/$NAME.o
echo "built build_ab/$NAME.so"
