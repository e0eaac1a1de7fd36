"""GPU: icelk_set_mask_polygon (k_polygon_mask) against the reference's masks (golden) and, at full frame size,
against the oracle; the rasterised mask then drives the detector exactly like an uploaded one."""
import numpy as np
import pytest

from test_oracle_mask import cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,poly,crop,want", cases(), ids=[c[0] for c in cases()])
def test_polygon_mask_matches_reference(ctx, name, poly, crop, want):
    ctx.set_mask_polygon(poly, crop[0], crop[1], want.shape[1], want.shape[0])
    got = ctx.download_mask()
    ctx.set_mask(None)
    assert np.array_equal(got, want)


def test_full_frame_fjord_equals_oracle_and_masks_the_detector(orc, synth):
    from iceberg_tracking_code_amd import Context
    w, h = 4000, 3000
    rng = np.random.default_rng(3)
    ang = np.sort(rng.uniform(0, 2 * np.pi, 200))
    rad = rng.uniform(700, 1400, 200)
    poly = np.stack([np.floor(2100 + 1.3 * rad * np.cos(ang)), np.floor(2600 + rad * np.sin(ang))], 1)
    c = Context(w, h, n_slots=1, max_pts=1 << 15)
    c.set_mask_polygon(poly, 100, 1000, w, h)
    got = c.download_mask()
    want = orc.polygon_mask(poly, 100, 1000, w, h)
    assert np.array_equal(got, want) and 0.2 < (got == 255).mean() < 0.9
    # the device-built mask and the same mask uploaded from the host select the same corners
    c.synth_frame(0, w, h, 0, 0, 1234)
    a = c.good_features(0, 5000, 0.007, 10, True, 10)
    c.set_mask(want)
    b = c.good_features(0, 5000, 0.007, 10, True, 10)
    frame = c.download_level(0, 0)
    c.close()
    assert len(a) == 5000 and np.array_equal(a, b)
    # ... and they are the oracle's corners under that mask, in order (s1:437 with mask=mask): the masked maximum, the
    # quality threshold derived from it and the candidates of the strip kernel at the full frame size
    ref = orc.good_features(frame, 5000, 0.007, 10, want, 10)
    assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(ref, np.float32).reshape(np.asarray(a).shape).view(np.uint8))
    xy = a.reshape(-1, 2).astype(int)
    assert np.all(want[xy[:, 1], xy[:, 0]] == 255)


def test_degenerate_polygons(ctx):
    ctx.set_mask_polygon(np.zeros((0, 2)), 0, 0, 33, 17)
    assert not ctx.download_mask().any()
    ctx.set_mask_polygon([(1, 1), (20, 9)], 0, 0, 33, 17)
    assert not ctx.download_mask().any()
    ctx.set_mask(None)
