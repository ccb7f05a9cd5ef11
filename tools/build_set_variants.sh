#!/bin/bash
# Measurement builds: libllamahip_<name>.so = the product objects with ONE translation unit (SRC, default gemv_set; e.g. SRC=decode)
# recompiled under extra flags (results of the *_ABLATE builds are wrong by construction).
# usage: [SRC=decode] tools/build_set_variants.sh name:"-DFLAG=.." ...
cd "$(dirname "$0")/../llama.swift_amd/csrc" || exit 1
SRC=${SRC:-gemv_set}
make -s -j8 all || exit 1
CXX="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden -Wall -Wno-unused-function -Wno-unused-result"
mkdir -p build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $CXX $flags -x hip -c -o build/variants/${SRC}_$name.o $SRC.hip || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o libllamahip_$name.so $(ls build/*.o | grep -v "/$SRC\.hip\.o") build/variants/${SRC}_$name.o || exit 1
  echo "built libllamahip_$name.so [$SRC: $flags]"
done
