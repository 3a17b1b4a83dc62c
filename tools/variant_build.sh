#!/bin/bash
# Build a VARIANT of the library next to the shipped one without touching it: one source recompiled with extra -D flags, linked
# with the regular objects into opticalflow_amd/csrc/build/var/libpwc_<name>.so (select it with PWC_HIP_LIB=<that path>).
# usage: tools/variant_build.sh <name> <source.hip> "<flags>"        e.g.  tools/variant_build.sh s1 pwc_stream3x3.hip "-DPWC_STREAM_EXP=1"
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/opticalflow_amd/csrc"
name="$1"; src="$2"; flags="$3"
make > /dev/null                                     # regular objects up to date
mkdir -p build/var
obj="build/var/${name}_${src%.hip}.o"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize $flags -c "$src" -o "$obj"
others=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "build/var/libpwc_${name}.so" $obj $others
echo "build/var/libpwc_${name}.so"
