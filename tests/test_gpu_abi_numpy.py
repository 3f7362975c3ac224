"""The C ABI is self-sufficient: a host with NumPy + ctypes only - no PyTorch in the process - uploads the
trial queue, runs mtmfft + coherence, sums over ranks with the library's own RCCL communicator and reproduces the
reference's vectors."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_numpy_only_host_reproduces_c1():
    from syncopy_amd import backend
    backend.require_gpu()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_numpy_only.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "abi numpy-only ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
