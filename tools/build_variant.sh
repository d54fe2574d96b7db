#!/bin/sh
# A tuning build of the library with extra -D flags: tools/build_variant.sh OUT.so [-DNAME=VALUE ...]
# (for tools/ab_libs.py: builds are only ever compared on one GPU box in one gpurun call)
out="$1"; shift
cd "$(dirname "$0")/.." && mkdir -p "$(dirname "$out")" && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off \
  -mllvm -disable-machine-licm -Xarch_host -mfma "$@" -Iinclude -Irimphony_amd/csrc \
  rimphony_amd/csrc/rimphony_hip.hip rimphony_amd/csrc/rimphony_diag.hip rimphony_amd/csrc/rimphony_group.hip rimphony_amd/csrc/rimphony_multi.hip -ldl -o "$out"
