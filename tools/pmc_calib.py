#!/usr/bin/env python3
"""Target of the FETCH_SIZE / WRITE_SIZE calibration passes (tools/collect_pmc.sh)."""
import ctypes as C
import sys
from pathlib import Path

import torch

lib = C.CDLL(str(Path(__file__).resolve().parent / "_diag" / "libpmc_calib.so"))
n = 1 << 28  # 2 GiB of fp64 / 1 GiB of fp32 per buffer
for elem, dt in ((8, torch.float64), (4, torch.float32)):
    src = torch.ones(n, dtype=dt, device="cuda")
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    for _ in range(3):
        lib.calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_int64(n), elem,
                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    del src, dst
print("calib bytes per launch: f64", n * 8, "f32", n * 4)
