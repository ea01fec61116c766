#!/bin/bash
# A variant of the whole library: ONE source recompiled with extra flags, linked with the default build's other objects.
#   tools/lib_variant.sh <name> <file.hip> [-DFLAG ...]   ->  tools/_trace/libcadre_<name>.so   (use: CADRE_HIP_LIB=<that> python tools/...)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
base=$(basename "$src" .hip)
mkdir -p tools/_trace
extra=$(python3 -c "
import sys; sys.path.insert(0, '.')
from cadre_amd import build
print(' '.join(build.EXTRA_FLAGS.get('$src', [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -c cadre_amd/csrc/$src -o tools/_trace/${base}_$name.o
objs=$(ls cadre_amd/csrc/build/default/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_trace/libcadre_$name.so $objs tools/_trace/${base}_$name.o
echo tools/_trace/libcadre_$name.so
