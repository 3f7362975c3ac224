"""The C-ABI library loads and exports every symbol include/spyhip.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "spyhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spyhip_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from syncopy_amd import _lib, build
    build.build()           # cross-compiles for gfx950 without a GPU
    return _lib.load()


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/spyhip.h but not exported by libspyhip.so"


def test_ctypes_table_matches_header(lib):
    from syncopy_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_string(lib):
    assert lib.spyhip_version() >= 100
    assert isinstance(lib.spyhip_last_error(), bytes)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import syncopy_amd as spy
    from syncopy_amd._lib import SpyHipError
    data = spy.synthdata.white_noise(nSamples=256, nChannels=2, nTrials=2, seed=1)
    with pytest.raises(SpyHipError):
        spy.freqanalysis(data, method="mtmfft", taper="hann")
    with pytest.raises(SpyHipError):
        spy.connectivityanalysis(data, method="coh", taper="hann")


def test_package_never_imports_the_oracle():
    """The product must not route through the CPU oracle or the kernel emulator."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "syncopy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+(oracle|spy_oracle|emu_driver|oracle_routines)\b", txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
                if "hip_emu.h" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_numpy_only_host_does_not_pull_torch_in():
    """`syncopy_amd.abi` (the NumPy + ctypes host of INTEGRATION.md) and the package itself import without PyTorch;
    on a box without a GPU the context creation fails loudly.  A NumPy-only host declares itself with SPY_NO_TORCH=1
    (otherwise _lib.load() imports an installed torch first, for the load order of libamdhip64 - ADVICE r2)."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['SPY_NO_TORCH'] = '1'\n"
            "import syncopy_amd as spy\n"
            "from syncopy_amd import abi\n"
            "from syncopy_amd.specest.tapers import taper_table\n"
            "assert 'torch' not in sys.modules, 'torch was imported'\n"
            "lib = abi._lib.load()\n"
            "assert lib.spyhip_version() >= 200\n"
            "assert 'torch' not in sys.modules\n"
            "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    # without the declaration the library loads torch BEFORE itself when torch is installed
    code2 = ("import sys; sys.path.insert(0, %r)\n"
             "from syncopy_amd import _lib\n"
             "assert 'torch' not in sys.modules\n"
             "_lib.load()\n"
             "assert 'torch' in sys.modules\n"
             "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
