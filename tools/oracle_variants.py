#!/usr/bin/env python3
"""How far apart are the build-dependent variants of OpenCV's arithmetic that the oracle names (icelk_oracle.c:
orc_set_variant)?  CPU only.  Runs the detector and the forward+backward tracker of BASELINE.json's C2 configuration and
of the reference's own parameters (REF, s1:240-248) on synthetic frames once per variant and compares with the default
(= what the HIP kernels compute):

    tracker:  max |dp| of p1 / p0r in px, count of status flips, count of `valid` flips (the FB decision of s1:333),
              features whose p1 moved by more than 1e-3 px (north_star's tolerance)
    detector: edit distance of the corner list (insertions + deletions against the default list, order-sensitive through
              the longest common subsequence of the first N), corners displaced in rank, size of the symmetric difference

    python tools/oracle_variants.py [--out profiles/r03_oracle_variants.json] [--quick]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from iceberg_tracking_code_amd import synth  # noqa: E402

DET = dict(qualityLevel=0.007, minDistance=10, blockSize=10)
CASES = {
    "C2": dict(w=4000, h=3000, maxCorners=10000, lk=dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01)), seed=1234),
    "REF": dict(w=3456, h=2304, maxCorners=0, lk=dict(winSize=(35, 35), maxLevel=4, criteria=(3, 25, 0.03)), seed=91),
}
LK_VARIANTS = [dict(lk_sums=1), dict(lk_sums=2)]
DET_VARIANTS = [dict(sobel_fma=1), dict(sobel_fma=2), dict(sobel_fma=3), dict(eig_fma=1), dict(sobel_fma=3, eig_fma=1)]


def lcs_len(a, b):
    """Longest common subsequence of two integer sequences (Hunt-Szymanski over positions; the lists are permutations of
    mostly the same keys, so this is n log n)."""
    import bisect
    pos = {}
    for j, v in enumerate(b):
        pos.setdefault(v, []).append(j)
    tails = []
    for v in a:
        for j in reversed(pos.get(v, ())):
            k = bisect.bisect_left(tails, j)
            if k == len(tails):
                tails.append(j)
            else:
                tails[k] = j
    return len(tails)


def corner_diff(ref, got):
    ka = (ref.reshape(-1, 2)[:, 1].astype(np.int64) << 16 | ref.reshape(-1, 2)[:, 0].astype(np.int64)).tolist()
    kb = (got.reshape(-1, 2)[:, 1].astype(np.int64) << 16 | got.reshape(-1, 2)[:, 0].astype(np.int64)).tolist()
    common = lcs_len(ka, kb)
    sa, sb = set(ka), set(kb)
    first = next((i for i, (x, y) in enumerate(zip(ka, kb)) if x != y), min(len(ka), len(kb)))
    return dict(n_default=len(ka), n_variant=len(kb), edit_distance=len(ka) + len(kb) - 2 * common,
                only_in_default=len(sa - sb), only_in_variant=len(sb - sa), first_difference_at_rank=first)


def track_diff(ref, got):
    d1 = np.abs(ref["p1"] - got["p1"]).max(axis=1)
    d0 = np.abs(ref["p0r"] - got["p0r"]).max(axis=1)
    both = (ref["st_fwd"] == 1) & (got["st_fwd"] == 1)
    return dict(features=int(len(d1)), max_dp1_px=float(d1[both].max()) if both.any() else 0.0,
                max_dp0r_px=float(d0[both & (ref["st_bwd"] == 1) & (got["st_bwd"] == 1)].max()),
                p1_bit_identical=int((d1 == 0).sum()), p1_moved_more_than_1e_3_px=int((d1[both] > 1e-3).sum()),
                p1_moved_more_than_1e_2_px=int((d1[both] > 1e-2).sum()),
                median_dp1_of_the_moved_px=float(np.median(d1[d1 > 0])) if (d1 > 0).any() else 0.0,
                status_flips_forward=int((ref["st_fwd"] != got["st_fwd"]).sum()),
                status_flips_backward=int((ref["st_bwd"] != got["st_bwd"]).sum()),
                valid_flips=int((ref["valid"] != got["valid"]).sum()),
                err_fwd_max_abs_diff=float(np.abs(ref["err_fwd"] - got["err_fwd"])[both].max()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_oracle_variants.json"))
    ap.add_argument("--quick", action="store_true", help="quarter-size frames (smoke run)")
    args = ap.parse_args()
    oracle.build()
    out = dict(note="default = exact int64 LK sums, separately rounded Sobel / eigenvalue arithmetic (what the HIP kernels "
                    "compute); variants = oracle/icelk_oracle.c orc_set_variant; synthetic frames (translation + <= 0.5 % "
                    "affine), detector s1:241-243", cases={})
    for name, c in CASES.items():
        w, h = (c["w"] // 4, c["h"] // 4) if args.quick else (c["w"], c["h"])
        sh, af = synth.shifts(2, seed=c["seed"]), synth.affines(2, seed=c["seed"])
        t0 = time.time()
        f = [synth.frame(w, h, int(sh[i, 0]), int(sh[i, 1]), c["seed"], affine=af[i]) for i in range(2)]
        det = lambda: oracle.good_features(f[0], c["maxCorners"], DET["qualityLevel"], DET["minDistance"], None, DET["blockSize"])  # noqa: E731
        base_c = det()
        pts = base_c.reshape(-1, 2)
        base_t = oracle.track_fb(f[0], f[1], pts, **c["lk"])
        res = dict(width=w, height=h, corners=int(len(pts)), lk=dict(winSize=list(c["lk"]["winSize"]), maxLevel=c["lk"]["maxLevel"],
                                                                      criteria=list(c["lk"]["criteria"])), tracker={}, detector={})
        for v in LK_VARIANTS:
            with oracle.variants(**v):
                res["tracker"][json.dumps(v, sort_keys=True)] = track_diff(base_t, oracle.track_fb(f[0], f[1], pts, **c["lk"]))
        for v in DET_VARIANTS:
            with oracle.variants(**v):
                e0 = oracle.min_eig_map(f[0], DET["blockSize"])
                got = det()
            e = oracle.min_eig_map(f[0], DET["blockSize"])
            d = corner_diff(base_c, got)
            d["eig_map_pixels_differing"] = int((e.view(np.uint32) != e0.view(np.uint32)).sum())
            d["eig_map_max_ulp"] = int(np.abs(e.view(np.int32).astype(np.int64) - e0.view(np.int32).astype(np.int64)).max())
            res["detector"][json.dumps(v, sort_keys=True)] = d
        res["seconds"] = round(time.time() - t0, 1)
        out["cases"][name] = res
        sys.stderr.write("%s done in %.0f s\n" % (name, res["seconds"]))
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
