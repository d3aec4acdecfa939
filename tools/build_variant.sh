#!/bin/bash
# Dev tool: builds a variant of the library with another generated window stream into exp/lib_<name>.so
# (MALCS_LIB=exp/lib_<name>.so python tools/time_window.py), then puts the default stream back.
#   tools/build_variant.sh noload GEN_EXP=noload
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p exp
env "$@" python tools/gen_diag_lds.py
cd algebra_amd/csrc
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -c malcs_diag.hip -o build/variant_diag.o
objs=$(ls build/*.o | grep -v -e variant_diag -e 'build/malcs_diag.o')
hipcc -shared -fPIC --offload-arch=gfx950 -o ../../exp/lib_$name.so $objs build/variant_diag.o -ldl
rm build/variant_diag.o
cd ../..
python tools/gen_diag_lds.py > /dev/null
echo "built exp/lib_$name.so"
