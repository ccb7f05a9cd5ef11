#!/bin/bash
# Measurement builds of the few-row kernel: libllamahip_<name>.so = the product objects with gemv_set.hip recompiled under extra flags
# (results of the LH_SET_ABLATE builds are wrong by construction).  usage: tools/build_set_variants.sh name:"-DFLAG=.." ...
cd "$(dirname "$0")/../llama.swift_amd/csrc" || exit 1
make -s -j8 all || exit 1
CXX="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden -Wall -Wno-unused-function -Wno-unused-result"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX $flags -x hip -c -o build/gemv_set_$name.o gemv_set.hip || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o libllamahip_$name.so $(ls build/*.o | grep -v "gemv_set") build/gemv_set_$name.o || exit 1
  echo "built libllamahip_$name.so [$flags]"
done
