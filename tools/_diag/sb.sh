for L in tools/_diag/libbase.so ilqr_iterative_tasks_amd/csrc/libi2lqr_hip.so; do I2LQR_LIB=$L python - <<PY
import os,sys
sys.path.insert(0,".")
from ilqr_iterative_tasks_amd import _abi
from pathlib import Path
_abi.LIB_PATH=Path(os.environ["I2LQR_LIB"]).resolve()
sys.argv=["x","tiled:f64:65536","tiled:f64:1048576","tiled:f32:65536","tiled:f64:16384","tiled:f64:4096"]
exec(open("tools/solve_bench.py").read())
PY
done
