"""Worker of tests/test_host_logic.py::test_two_rank_gloo_gather (launched by torch.distributed.run, CPU, gloo)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from iceberg_tracking_code_amd import sharding, synth  # noqa: E402
from reference_loops import OracleCv, run_reference_loop  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
track_len, n_frames = 2, 9
shifts = synth.shifts(n_frames, seed=21, max_step_px=2.0)
fp = dict(maxCorners=80, qualityLevel=0.01, minDistance=8, blockSize=5)
lk = dict(winSize=(21, 21), maxLevel=2, criteria=(3, 30, 0.01))


def segments_for(f0, f1):
    frames = [synth.frame(200, 150, int(shifts[i, 0]), int(shifts[i, 1]), 21) for i in range(f0, f1)]
    segs = run_reference_loop(frames, track_len, fp, lk, cv=OracleCv(oracle))
    return [np.float32(t).reshape(-1, track_len + 1, 2) for _, t, _ in segs]


import torch  # noqa: E402

f0, f1 = sharding.frame_block(n_frames, track_len, rank, world)
mine = segments_for(f0, f1)
allc = sharding.gather_counts([len(t) for t in mine], dist)
# the padded all_gather of the track tables (BASELINE.json configs[3]; s1:394-395 arrays): R rows per segment
R = fp["maxCorners"]
tab = torch.zeros((len(mine), R, track_len + 1, 2), dtype=torch.float32)
cnt = torch.zeros(len(mine), dtype=torch.int32)
for s, t in enumerate(mine):
    tab[s, :len(t)] = torch.from_numpy(t)
    cnt[s] = len(t)
allt = sharding.gather_tables(tab, cnt, dist)
if rank == 0:
    whole = segments_for(0, n_frames)
    assert len(allc) == sharding.segment_count(n_frames, track_len) == len(whole), (allc, whole)
    assert np.array_equal(allc, [len(t) for t in whole]), (allc, whole)
    assert len(allt) == len(whole)
    for (t, n), w in zip(allt, whole):
        assert n == len(w) and np.array_equal(t, w)
    print("GLOO_GATHER_OK", allc.tolist())
dist.barrier()
dist.destroy_process_group()
