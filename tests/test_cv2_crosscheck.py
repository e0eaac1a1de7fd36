"""Opportunistic cross-check of the oracle against a real OpenCV.

OpenCV is not installed in the development image nor on the GPU box, so these tests are SKIPPED there and the oracle
stays "parity unpinned" for the tracking path.  Wherever `import cv2` works they state the
agreement BASELINE.json's north_star asks for: displacements within 1e-3 px, status identical, corner sets identical,
integer stages exact (gray may differ by 1 LSB between the OpenCV 3.x and 4.x coefficient sets)."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="OpenCV not installed: the tracking-path oracle stays unpinned")


def test_versions_are_reported():
    print("cv2", cv2.__version__, "threads", cv2.getNumThreads())


def test_gray_and_pyrdown(orc, synth):
    rgb = synth.rgb_from_gray_seeded(321, 200, 10, -20, 8)
    ref = cv2.cvtColor(rgb, cv2.COLOR_BGR2GRAY)
    a3, a4 = orc.bgr2gray(rgb, 3), orc.bgr2gray(rgb, 4)
    assert np.array_equal(ref, a3) or np.array_equal(ref, a4)
    g = synth.frame(403, 301, 0, 0, 5)
    assert np.array_equal(cv2.pyrDown(g), orc.pyrdown(g))


def test_good_features(orc, synth):
    g = synth.frame(640, 480, 0, 0, 7)
    for kw in (dict(maxCorners=200, qualityLevel=0.007, minDistance=10, blockSize=10),
               dict(maxCorners=0, qualityLevel=0.01, minDistance=7, blockSize=3)):
        ref = cv2.goodFeaturesToTrack(g, mask=None, **kw)
        got = orc.good_features(g, kw["maxCorners"], kw["qualityLevel"], kw["minDistance"], None, kw["blockSize"])
        assert ref is not None and got is not None
        assert set(map(tuple, ref.reshape(-1, 2))) == set(map(tuple, got.reshape(-1, 2)))
        assert np.array_equal(ref, got)      # and in the same order


def test_pyrlk(orc, synth):
    a, b = synth.frame(640, 480, 0, 0, 5), synth.frame(640, 480, 300, -200, 5)
    p0 = cv2.goodFeaturesToTrack(a, maxCorners=300, qualityLevel=0.007, minDistance=10, blockSize=10)
    for lk in (dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01)),
               dict(winSize=(35, 35), maxLevel=4, criteria=(3, 25, 0.03))):
        p1, st, err = cv2.calcOpticalFlowPyrLK(a, b, p0, None, **lk)
        q1, qs, qe = orc.pyrlk(a, b, p0, None, **lk)
        assert np.array_equal(st, qs)
        ok = st.ravel() == 1
        assert np.abs(p1.reshape(-1, 2)[ok] - q1.reshape(-1, 2)[ok]).max() < 1e-3
        assert np.abs(err.ravel()[ok] - qe.ravel()[ok]).max() < 1e-3
