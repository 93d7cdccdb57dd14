#!/bin/bash
# Diagnostic build with in-kernel phase stamps (never shipped, never timed): libi2lqr_stamps.so.
# A translation unit that does not compile with the stamps (the wave / lane kernels of i2lqr_abi.hip
# trip a code generator assertion on some compiler builds) falls back to the product object, so the
# stamps of the other kernel families stay usable.
cd "$(dirname "$0")/../ilqr_iterative_tasks_amd/csrc" || exit 1
make -s >/dev/null 2>&1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DI2LQR_STAMPS ${EXTRA:-}"  # EXTRA: experiment switches
objs=""
for tu in abi quad group lane12 lane12qr lane12f lanepair; do
  ( /opt/rocm/bin/hipcc $FLAGS -c -o /tmp/i2lqr_${tu}_st.o i2lqr_${tu}.hip >/tmp/i2lqr_${tu}_st.log 2>&1 \
      || { echo "stamps: i2lqr_${tu}.hip does not build with -DI2LQR_STAMPS, using the product object"; \
           cp _obj/i2lqr_${tu}.o /tmp/i2lqr_${tu}_st.o; } ) &
  objs="$objs /tmp/i2lqr_${tu}_st.o"
done
wait
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libi2lqr_stamps.so $objs
mkdir -p ../../tools/_diag && cp /tmp/libi2lqr_stamps.so ../../tools/_diag/libi2lqr_stamps.so
ls -la ../../tools/_diag/libi2lqr_stamps.so
