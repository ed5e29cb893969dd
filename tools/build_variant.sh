#!/bin/bash
# Development aid: a variant of libpygat_amd.so with ONE source recompiled under extra flags (diagnostic macros, experiments),
# linked with the standard objects:   tools/build_variant.sh <name> <source.hip> "<extra hipcc flags>"
# -> pygat_amd/libpygat_amd_<name>.so ; run with PYGAT_AMD_LIB=<that path> (pygat_amd/_lib.py).  Variants are never shipped:
# *.so is git-ignored, and the default library is the only one the package loads unless that variable is set.
set -e
NAME=$1; SRC=$2; EXTRA=$3
D=$(cd "$(dirname "$0")/../pygat_amd/csrc" && pwd)
make -C "$D" -j8 > /dev/null
mkdir -p "$D/build/variants"
OBJ="$D/build/variants/${NAME}_$(basename "$SRC" .hip).o"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EXTRA -c "$D/$SRC" -o "$OBJ"
OBJS=$(ls "$D"/build/*.o | grep -v "/$(basename "$SRC" .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/../libpygat_amd_${NAME}.so" $OBJS "$OBJ"
echo "built $D/../libpygat_amd_${NAME}.so"
