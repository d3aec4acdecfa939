#!/bin/bash
# Dev tool: builds build_tools/libmalcs_<name>.so for generator settings given as "name:ENV=val,ENV=val" arguments,
# then restores the default stream.  Use with MALCS_LIB=build_tools/libmalcs_<name>.so.
set -e
cd "$(dirname "$0")/.."
make -C algebra_amd/csrc -j8 -s
mkdir -p build_tools
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for kv in $envs; do export "$kv"; done; python tools/gen_diag_lds.py )
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -c algebra_amd/csrc/malcs_diag.hip -o build_tools/diag_$name.o
  objs=$(ls algebra_amd/csrc/build/*.o | grep -v malcs_diag.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o build_tools/libmalcs_$name.so $objs build_tools/diag_$name.o -ldl
  rm build_tools/diag_$name.o
done
python tools/gen_diag_lds.py
make -C algebra_amd/csrc -j8 -s
