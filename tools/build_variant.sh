#!/bin/bash
# A/B builds of the quad12 lane kernel: recompiles ONLY i2lqr_lane12.hip with extra -D switches and
# links it with the product objects of the other translation units.
#   tools/build_variant.sh <name> [-DSWITCH=1 ...]   ->  tools/_diag/lib<name>.so
# (never shipped; tools/ab_bench.py takes the path as the 4th field of a variant)
set -e
name="$1"; shift
cd "$(dirname "$0")/../ilqr_iterative_tasks_amd/csrc"
mkdir -p ../../tools/_diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/var_${name}_lane12.o i2lqr_lane12.hip 2>&1 \
  | grep -A12 "k_lane_iterate_rowsIdNS_6Quad12IdEELb1" | grep -E "VGPRs|AGPRs|Spill|ScratchSize" | head -6
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/_diag/lib${name}.so \
  _obj/i2lqr_abi.o _obj/i2lqr_group.o _obj/i2lqr_quad.o /tmp/var_${name}_lane12.o
ls -la ../../tools/_diag/lib${name}.so
