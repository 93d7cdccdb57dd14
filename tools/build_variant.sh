#!/bin/bash
# A/B builds of ONE translation unit (default: the quad12 lane kernels, i2lqr_lane12.hip; TU=lanepair
# for the helper-wavefront kernels, ...): recompiles it with extra -D switches and links it with the
# product objects of the others.
#   [TU=lanepair] tools/build_variant.sh <name> [-DSWITCH=1 ...]   ->  tools/_diag/lib<name>.so
# (never shipped; tools/ab_bench.py takes the path as the 4th field of a variant)
set -e
name="$1"; shift
tu="${TU:-lane12}"
cd "$(dirname "$0")/../ilqr_iterative_tasks_amd/csrc"
mkdir -p ../../tools/_diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" \
  -c -o /tmp/var_${name}_${tu}.o i2lqr_${tu}.hip
objs=""
for t in abi group quad lane12 lane12qr lane12f lanepair; do
  if [ "$t" = "$tu" ]; then objs="$objs /tmp/var_${name}_${tu}.o"; else objs="$objs _obj/i2lqr_${t}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/_diag/lib${name}.so $objs
ls -la ../../tools/_diag/lib${name}.so
