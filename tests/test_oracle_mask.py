"""CPU: oracle/mask_oracle.c against masks the REFERENCE produced (tests/golden/make_mask_golden.py ran
Camera.mask_meshgrid / matplotlib contains_points as s1:285-291 does) -- byte for byte, including pixel centres that
lie exactly on edges and vertices of the integer polygons."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mask_golden.npz")


def cases():
    z = np.load(GOLD, allow_pickle=False)
    return [(str(n), z[str(n) + "_poly"], z[str(n) + "_crop"], z[str(n) + "_mask"]) for n in z["names"]]


@pytest.mark.parametrize("name,poly,crop,want", cases(), ids=[c[0] for c in cases()])
def test_polygon_mask_matches_reference(orc, name, poly, crop, want):
    got = orc.polygon_mask(poly, crop[0], crop[1], want.shape[1], want.shape[0])
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    assert set(np.unique(got)) <= {0, 255}
