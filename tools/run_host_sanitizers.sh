#!/bin/bash
# The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer: builds
# libi2lqr_hip_asan.so (make -C ilqr_iterative_tasks_amd/csrc asan) and runs the tests that drive
# the real library through ctypes with it: every export's validation path (NULL handles / buffers),
# the layout recommendation over the measured table, workspace sizes, the handle registry
# (destroy of NULL / foreign / stale pointers), the error string, the RCCL binding's availability
# check.  Run on a box WITHOUT a GPU: with one visible, AMD's ASan runtime intercepts
# hsa_amd_memory_pool_allocate for its device-side shadow and aborts on this pool ("out of memory:
# allocator is trying to allocate 0x400000 bytes" — GPU ASan is not available here).  The paths
# behind i2lqr_create (launchers, chunk scheduler) are reached through the DRY-RUN handle of this
# build (I2LQR_DRY_RUN=1: no device, launches become checked records; second half below).  The
# interpreter is not instrumented: the ASan runtime is preloaded, leak detection is off (CPython's
# arenas).
set -e
cd "$(dirname "$0")/.."
make -s -C ilqr_iterative_tasks_amd/csrc asan
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 \
       UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
       I2LQR_LIB_PATH=$PWD/ilqr_iterative_tasks_amd/csrc/libi2lqr_hip_asan.so
python -m pytest tests/test_abi.py tests/test_layout_table.py -q "$@"
# ... and the host code BEHIND a live handle (round 6): dry-run launches (csrc/i2lqr_dryrun.hpp) over
# random configurations — workspace carving, the chunked solve's scheduler, LDS budgeting, the
# sharded round — every pointer a kernel would be handed checked against the declared ranges
I2LQR_DRY_RUN=1 python tools/dry_run_fuzz.py --configs ${DRY_RUN_CONFIGS:-1000}
