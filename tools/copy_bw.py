#!/usr/bin/env python3
"""Bandwidth of the plain grid-stride streaming copy of tools/pmc_calib.hip at 8 B and 4 B per lane
(read + write): a reference point for the HBM rates of the lane kernels (MI355X: 4.4-4.6 TB/s for
both widths; build the library with tools/collect_pmc.sh or hipcc -shared tools/pmc_calib.hip)."""
import ctypes as C, sys
from pathlib import Path
import torch
lib = C.CDLL(str(Path(__file__).resolve().parent / "_diag" / "libpmc_calib.so"))
n = 1 << 28
for elem, dt in ((8, torch.float64), (4, torch.float32)):
    src = torch.ones(n, dtype=dt, device="cuda"); dst = torch.empty_like(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(3):
        e0.record()
        lib.calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_int64(n), elem, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{elem} B/lane copy: {2*n*elem/ms/1e6:.0f} GB/s (read+write)")
    del src, dst
