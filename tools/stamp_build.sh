#!/bin/bash
# Diagnostic build with in-kernel phase stamps (never shipped, never timed): libi2lqr_stamps.so
set -e
cd "$(dirname "$0")/../ilqr_iterative_tasks_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DI2LQR_STAMPS"
/opt/rocm/bin/hipcc $FLAGS -c -o /tmp/i2lqr_abi_st.o i2lqr_abi.hip &
/opt/rocm/bin/hipcc $FLAGS -c -o /tmp/i2lqr_quad_st.o i2lqr_quad.hip &
/opt/rocm/bin/hipcc $FLAGS -c -o /tmp/i2lqr_group_st.o i2lqr_group.hip
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libi2lqr_stamps.so /tmp/i2lqr_abi_st.o /tmp/i2lqr_group_st.o /tmp/i2lqr_quad_st.o
mkdir -p ../../tools/_diag && cp /tmp/libi2lqr_stamps.so ../../tools/_diag/libi2lqr_stamps.so
