set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
SPY_FUZZ_OFFSET=500000 python tools/fuzz_diag.py conn 23 284 2>&1 | grep -v Warning | tee gpurun_out/r6e/diag_500000.log
SPY_FUZZ_OFFSET=900000 python tools/fuzz_diag.py conn 668 2>&1 | grep -v Warning | tee gpurun_out/r6e/diag_900000_conn.log
SPY_FUZZ_OFFSET=900000 python tools/fuzz_diag.py mtmfft 1624 2>&1 | grep -v Warning | tee gpurun_out/r6e/diag_900000_mtmfft.log
SPY_FUZZ_OFFSET=500000 python - <<'PY' 2>&1 | grep -v Warning | tee gpurun_out/r6e/diag_284.log
import os, sys
import numpy as np
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import syncopy_amd as spy
import test_gpu_fuzz as T
from oracle_routines import ORACLE_CONN
seed = 284
rng = np.random.default_rng(6000 + T.OFFSET + seed)
# replay test_connectivity_random_options up to the call
import inspect
src = inspect.getsource(T.test_connectivity_random_options)
print(src[:1500])
PY
SPY_FUZZ_OFFSET=500000 timeout 600 python -m pytest "tests/test_gpu_fuzz.py::test_granger_random_networks[94]" -x -q 2>&1 | tail -30 | tee gpurun_out/r6e/granger94.log
