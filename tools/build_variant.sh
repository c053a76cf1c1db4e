#!/bin/bash
# Build an A/B variant of libkws_hip.so with extra compiler flags into tools/bin/ (git-ignored, shipped by gpurun):
#   tools/build_variant.sh <name> [-DMACRO ...]        -> tools/bin/libkws_<name>.so
# Select it at run time with KWS_HIP_LIB=tools/bin/libkws_<name>.so (kws/_native/__init__.py).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/keyword-spotting_amd/csrc
OBJ=$ROOT/tools/bin/obj_$NAME
mkdir -p "$OBJ"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -Wall -Wno-unused-function -fvisibility=hidden -ffp-contract=off -DKWS_BUILD $*"
for f in kws_api kws_ingest kws_dsblock kws_mfcc kws_mfcc_f64 kws_dscnn kws_cnntrad; do
  # only the translation units a macro can touch are rebuilt per variant; the others are linked from the main build
  if [ "$f" != kws_api ] && [ -f "$SRC/build/$f.o" ] && ! grep -q "KWS_MFCC_\|KWS_DSCNN_\|KWS_X_\|kws_mfcc_dev.h" "$SRC/$f.hip"; then cp "$SRC/build/$f.o" "$OBJ/$f.o"; continue; fi
  /opt/rocm/bin/hipcc $FLAGS -c "$SRC/$f.hip" -o "$OBJ/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$ROOT/tools/bin/libkws_$NAME.so" "$OBJ"/*.o
echo "built tools/bin/libkws_$NAME.so"
