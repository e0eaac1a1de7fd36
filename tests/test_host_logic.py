"""Host-side logic (CPU only): wire format, loop bookkeeping, synthetic data, sharding."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleCv:
    def __init__(self, orc):
        self.o = orc

    def calcOpticalFlowPyrLK(self, a, b, p0, p1, **kw):
        return self.o.pyrlk(a, b, p0, p1, **kw)

    def goodFeaturesToTrack(self, img, mask=None, **kw):
        return self.o.good_features(img, kw["maxCorners"], kw["qualityLevel"], kw["minDistance"], mask,
                                    kw.get("blockSize", 3))


def test_npz_name_is_what_s2_parses(tmp_path):
    """s2_cam_to_utm.py:177 takes split('_')[-2].split('sec')[0] as the tracking interval and :197-198
    split('_')[0] as the timestamp; s1:394 builds '<image without extension>_{T*dt}sec_at_{dt}sec_tracks.npz'."""
    from iceberg_tracking_code_amd import npz_name, save_tracks
    name = npz_name(str(tmp_path / "20190724-101500.jpg"), 2, 60)
    base = os.path.basename(name)
    assert base == "20190724-101500_120sec_at_60sec_tracks.npz"
    assert int(base.split("_")[-2].split("sec")[0]) == 60
    assert base.split("_")[0] == "20190724-101500"
    tracks = np.zeros((5, 3, 2), np.float32)
    quality = np.zeros((5, 2), np.float32)
    save_tracks(name, tracks, quality)
    z = np.load(name)
    assert z["tracks"].shape == (5, 3, 2) and z["trackquality"].shape == (5, 2)
    assert np.asarray(z["tracks"].tolist()).shape == (5, 3, 2)     # s2:233-234 does .tolist()


def test_segment_time_rule():
    from iceberg_tracking_code_amd import segment_time_ok
    ok = ["20190724-101500.jpg", "20190724-101601.jpg", "20190724-101659.jpg"]
    assert segment_time_ok(ok, 60)
    assert not segment_time_ok(["20190724-101500.jpg", "20190724-101800.jpg"], 60)
    assert not segment_time_ok(["20190724-101500.jpg", "20190724-101557.jpg"], 60)   # -3 s


@pytest.mark.parametrize("track_len", [1, 2, 3])
def test_reference_loop_bookkeeping(orc, synth, track_len):
    from reference_loops import run_reference_loop
    frames, sh = synth.sequence(240, 180, 7, seed=5, max_step_px=2.0)
    fp = dict(maxCorners=120, qualityLevel=0.01, minDistance=8, blockSize=5)
    lk = dict(winSize=(21, 21), maxLevel=2, criteria=(3, 30, 0.01))
    segs = run_reference_loop(frames, track_len, fp, lk, cv=OracleCv(orc))
    assert len(segs) == (len(frames) - 1) // track_len
    for first, tracks, quality in segs:
        assert first % track_len == 0
        t = np.asarray(tracks, np.float32)
        q = np.asarray(quality, np.float32)
        assert t.ndim == 3 and t.shape[1:] == (track_len + 1, 2) and q.shape == (len(t), track_len)
        assert len(t) > 50 and np.all(q < 1.0)
        # every vertex step is the known frame-to-frame translation
        for v in range(track_len):
            d = t[:, v + 1] - t[:, v]
            flow = synth.true_flow(sh[first + v], sh[first + v + 1])
            assert np.median(np.abs(d - flow)) < 0.05


def test_demo_class_counts(orc, synth):
    """s0_1-shaped LucasKanade: BASELINE.json configs[0] plumbing (2 frames 640x480, 200 corners)."""
    from reference_loops import LucasKanade
    frames, _ = synth.sequence(640, 480, 2, seed=1234)
    lk = LucasKanade(frames, detect_interval=3, time_spacing=120, cv=OracleCv(orc),
                     feature_params=dict(maxCorners=200, qualityLevel=0.007, minDistance=10, blockSize=10))
    tracks = lk.run()
    assert lk.track_counts == [0]                 # one detection, on frame 0 (s0_1:129 prints before detecting)
    assert 150 <= len(tracks) <= 200 and all(len(t) == 2 for t in tracks)


def test_synth_is_deterministic_and_moves_the_right_way(synth):
    a = synth.frame(200, 100, 0, 0, 3)
    assert np.array_equal(a, synth.frame(200, 100, 0, 0, 3))
    assert a.std() > 15 and 0 < a.min() and a.max() < 255
    # a shift of exactly 5 px: frame(x) = texture(x + 5)  ->  content moves by -5
    b = synth.frame(200, 100, 5 * 256, 0, 3)
    assert np.array_equal(b[:, :-5], a[:, 5:])
    assert np.allclose(synth.true_flow((0, 0), (5 * 256, 0)), (-5, 0))
    band = synth.frame(200, 100, 77, -33, 3, rows=(40, 60))
    assert np.array_equal(band, synth.frame(200, 100, 77, -33, 3)[40:60])
    s = synth.shifts(10, seed=1)
    assert s.shape == (10, 2) and np.all(s[0] == 0) and np.abs(np.diff(s, axis=0)).max() <= 3 * 256


def test_shard_plan():
    from iceberg_tracking_code_amd import sharding as sh
    assert sh.segment_count(86400, 2) == 43199 and sh.segment_count(2, 2) == 0 and sh.segment_count(3, 2) == 1
    for world in (1, 2, 3, 8):
        blocks = [sh.segment_block(43199, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == 43199
        assert all(b[1] == n[0] for b, n in zip(blocks, blocks[1:]))
        assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1
    # frames: a rank's block plus the closing frame (shared with the next rank)
    assert sh.frame_block(9, 2, 0, 2) == (0, 5) and sh.frame_block(9, 2, 1, 2) == (4, 9)
    assert np.array_equal(sh.gather_counts([3, 4, 5]), [3, 4, 5])


def test_two_rank_gloo_gather():
    """world_size 2 on CPU (gloo): each rank tracks its own segment block with the oracle, counts are gathered."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "tests", "gloo_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GLOO_GATHER_OK" in out.stdout


def test_bench_helpers():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.ping_pong(4, 9) == [0, 1, 2, 3, 2, 1, 0, 1, 2]
    assert bench.top_level_of(4000, 3000, (21, 21), 3) == 3
    # 18.7 MB per LK call at C2 (N = 10 000, 21x21, maxLevel 3), 19.7 MB pyramid per frame
    assert abs(bench.lk_algorithmic_bytes(4000, 3000, (21, 21), 3, 10000) / 1e6 - 18.7) < 0.2
    assert abs(bench.pyramid_algorithmic_bytes(4000, 3000, 3) / 1e6 - 19.7) < 0.1
    numa = bench.numa_placement(0)      # diagnostics of the PCIe leg: never raises, GPU or not
    assert set(numa) >= {"cpu", "cpu_node", "gpu_node"}


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` as the driver invokes it (no launcher, no WORLD_SIZE): the parent starts N child processes
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set before anything touches a GPU, relays rank 0's line and passes a
    failing rank's exit code on.  (Echo mode: the children report what they were handed and exit before importing torch.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["ICELK_BENCH_SPAWN_ECHO"] = str(tmp_path)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "8", "--warmup", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["RANK"] == "0" and line["WORLD_SIZE"] == "3" and line["gpus"] == 3 and line["steps"] == 8
    ranks = [json.load(open(tmp_path / ("rank_%d.json" % r))) for r in range(3)]
    assert [r["RANK"] for r in ranks] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in ranks] == ["0", "1", "2"]
    assert len({r["MASTER_PORT"] for r in ranks}) == 1 and all(r["MASTER_ADDR"] == "127.0.0.1" for r in ranks)
    env["ICELK_BENCH_SPAWN_ECHO_FAIL"] = "7"
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 7 and "ranks exited non-zero" in out.stderr
    # one rank per GPU of a shared device (tools/two_rank.sh): every rank is handed device 0
    env.pop("ICELK_BENCH_SPAWN_ECHO_FAIL")
    env["ICELK_BENCH_SHARED_DEVICE"] = "1"
    subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    assert [json.load(open(tmp_path / ("rank_%d.json" % r)))["LOCAL_RANK"] for r in range(3)] == ["0", "0", "0"]
