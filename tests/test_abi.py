"""C-ABI library: loads, exports every symbol include/icelk.h declares, fails loudly without a GPU, and the
product never touches the oracle."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "icelk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(icelk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from iceberg_tracking_code_amd import _lib
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes table and the header describe the same surface
    assert sorted(_lib.SIGNATURES) == names
    assert lib.icelk_version() >= 100
    assert lib.icelk_prof_count() > 5 and lib.icelk_prof_name(3) == b"lk_fb"


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from iceberg_tracking_code_amd import Context, IcelkError, _lib
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.icelk_create(0, 640, 480, 2, 1024, C.byref(h))
    assert rc == _lib.EHIP and not h.value
    assert b"HIP device" in lib.icelk_last_error(None)
    with pytest.raises(IcelkError):
        Context(640, 480)
    import numpy as np
    import iceberg_tracking_code_amd as cv
    with pytest.raises(IcelkError):
        cv.goodFeaturesToTrack(np.zeros((48, 64), np.uint8), 10, 0.01, 5)


def test_bad_arguments_are_rejected_before_the_gpu():
    from iceberg_tracking_code_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.icelk_create(0, 0, 480, 2, 1024, C.byref(h)) == _lib.EARG
    assert lib.icelk_create(0, 70000, 480, 2, 1024, C.byref(h)) == _lib.EARG
    assert lib.icelk_destroy(None) == _lib.EARG
    assert lib.icelk_sync(None) == _lib.EARG
    import numpy as np
    import iceberg_tracking_code_amd as cv
    with pytest.raises(ValueError):
        cv.goodFeaturesToTrack(np.zeros((48, 64), np.float32), 10, 0.01, 5)
    with pytest.raises(ValueError):
        cv.calcOpticalFlowPyrLK(np.zeros((48, 64), np.uint8), np.zeros((40, 64), np.uint8), np.zeros((1, 1, 2), np.float32))
    with pytest.raises(ValueError):
        cv.cvtColor(np.zeros((4, 4, 3), np.uint8), 99)
    # N == 0 is not an error and needs no GPU (cv2 returns empty outputs; the reference guards with len(tracks) > 0)
    p, st, er = cv.calcOpticalFlowPyrLK(np.zeros((48, 64), np.uint8), np.zeros((48, 64), np.uint8),
                                        np.zeros((0, 1, 2), np.float32))
    assert p.shape == (0, 1, 2) and st.shape == (0, 1) and er.shape == (0, 1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "iceberg_tracking_code_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "libicelk_oracle" not in text and "icelk_oracle.c" not in text.replace("oracle/icelk_oracle.c", ""), f
    # bench.py may use it only inside cpu_baseline()
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"^\s*import oracle", bench, flags=re.M)]
    start = bench.index("def cpu_baseline"), bench.index("def main")
    assert uses and all(start[0] < u < start[1] for u in uses)


def test_cv2_constants():
    import iceberg_tracking_code_amd as cv
    assert (cv.TERM_CRITERIA_COUNT, cv.TERM_CRITERIA_EPS, cv.COLOR_BGR2GRAY) == (1, 2, 6)
    assert cv.REF_LK_PARAMS["winSize"] == (35, 35) and cv.REF_LK_PARAMS["maxLevel"] == 4
    assert cv.REF_LK_PARAMS["criteria"] == (3, 25, 0.03)
    assert cv.REF_FEATURE_PARAMS == dict(maxCorners=50000000, qualityLevel=0.007, minDistance=10, blockSize=10)


def test_header_is_plain_c(tmp_path):
    """include/icelk.h is a C header: a C11 caller (tests/abi_smoke.c) compiles against it with gcc, warnings as
    errors, and links against the shared library without any C++ or torch symbol in the way."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    obj = tmp_path / "abi_smoke.o"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-c",
                           os.path.join(here, "abi_smoke.c"), "-o", str(obj)])
    assert obj.stat().st_size > 0
