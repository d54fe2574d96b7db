#!/bin/sh
# diagnostic build with region timers (see tools/region_profile.py, tools/region_profile_group.py)
# (-DRIM_GROUP_WAVES=4: the timers live in 256 B of dynamic LDS, which takes the group kernel's 7.6 KB per wave to the
# 8 KB allocation granule -- 20 waves per CU would need exactly 160 KB and are then not all resident)
cd "$(dirname "$0")/.." && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -disable-machine-licm \
  -DRIM_PROF -DRIM_COOP_DIAG -DRIM_GROUP_WAVES=4 -Iinclude -Irimphony_amd/csrc rimphony_amd/csrc/rimphony_hip.hip rimphony_amd/csrc/rimphony_diag.hip rimphony_amd/csrc/rimphony_group.hip rimphony_amd/csrc/rimphony_multi.hip -ldl -o rimphony_amd/librimphony_prof.so
# the same with only the cooperative-tail counters (tools/ab_assist.py):
cd "$(dirname "$0")/.." && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -disable-machine-licm \
  -DRIM_COOP_DIAG -Iinclude -Irimphony_amd/csrc rimphony_amd/csrc/rimphony_hip.hip rimphony_amd/csrc/rimphony_diag.hip rimphony_amd/csrc/rimphony_group.hip rimphony_amd/csrc/rimphony_multi.hip -ldl -o rimphony_amd/librimphony_diag.so
# execution counters instead of timers (tools/hit_profile.py):
cd "$(dirname "$0")/.." && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -disable-machine-licm \
  -DRIM_PROF -DRIM_PROF_COUNTS -DRIM_GROUP_WAVES=4 -Iinclude -Irimphony_amd/csrc rimphony_amd/csrc/rimphony_hip.hip rimphony_amd/csrc/rimphony_diag.hip rimphony_amd/csrc/rimphony_group.hip rimphony_amd/csrc/rimphony_multi.hip -ldl -o rimphony_amd/librimphony_hits.so
