#!/bin/bash
# Diagnostic build with in-kernel phase stamps (never shipped, never timed): libi2lqr_stamps.so
set -e
cd "$(dirname "$0")/../ilqr_iterative_tasks_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DI2LQR_STAMPS \
  -shared -o /tmp/libi2lqr_stamps.so i2lqr_abi.hip
mkdir -p ../../tools/_diag && cp /tmp/libi2lqr_stamps.so ../../tools/_diag/libi2lqr_stamps.so
