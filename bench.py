#!/usr/bin/env python3
"""bench.py -- frame-pairs/s and tracked-features/s of the sparse LK tracking loop on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the reference's frame loop body (s1_lucaskanade_tracking.py:307-450) over one new
4000x3000 frame that is already resident in HBM: Gaussian pyramid of the new frame, fused forward +
backward pyramidal LK of all live features against the previous frame, forward-backward filter and
track-table append, and -- every `track_len` (= 2, s1:128) frames -- Shi-Tomasi detection of up to 10 000
new features (the reference's detector parameters s1:240-243 with maxCorners capped at 10 000 as
BASELINE.json configs[1] asks).  Nothing is skipped or cached inside the timed region; no host transfer
of images is in it either (the PCIe-inclusive rate is in DESIGN.md).

Multi-GPU: every rank runs its own shard of independent segments (weak scaling, no data-path collective);
the only collective is the final RCCL all-gather of the per-rank feature counts (BASELINE.json
north_star); barriers and the max over ranks of the elapsed time go over gloo.  rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1]: single MI355X, 4000x3000 gray pair, 10k Shi-Tomasi features, 21x21, 3 levels
    "c2": dict(w=4000, h=3000, max_corners=10000, win=(21, 21), max_level=3, criteria=(3, 30, 0.01),
               name="C2: 4000x3000 gray, 10k Shi-Tomasi features, winSize 21x21, maxLevel 3 (4 pyramid images), "
                    "criteria (30, 0.01), track_len 2"),
    # BASELINE.json configs[4]
    "c5": dict(w=5760, h=3840, max_corners=50000, win=(31, 31), max_level=5, criteria=(3, 30, 0.01),
               name="C5: 5760x3840 gray, 50k features, winSize 31x31, maxLevel 5, track_len 2"),
    # the reference's own literals (s1:240-248) on its typical frame size (create_calibration_file.py:18)
    "ref": dict(w=3456, h=2304, max_corners=0, win=(35, 35), max_level=4, criteria=(3, 25, 0.03),
                name="REF: 3456x2304 gray, uncapped features, winSize 35x35, maxLevel 4, criteria (25, 0.03)"),
}
DETECT = dict(qualityLevel=0.007, minDistance=10, blockSize=10)   # s1:241-243
TRACK_LEN = 2                                                       # s1:128
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def lk_algorithmic_bytes(w, h, win, top_level, n):
    """compulsory HBM bytes of one LK call (derivatives on the fly)."""
    total = 0.0
    lw, lh = w, h
    for _ in range(top_level + 1):
        total += min(n * ((win[0] + 3) * (win[1] + 3) + (win[0] + 1) * (win[1] + 1)), 2.0 * lw * lh)
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
    return total + 21.0 * n


def pyramid_algorithmic_bytes(w, h, top_level):
    total, lw, lh = 0.0, w, h
    for _ in range(top_level):
        nw, nh = (lw + 1) // 2, (lh + 1) // 2
        total += lw * lh + nw * nh
        lw, lh = nw, nh
    return total


def top_level_of(w, h, win, max_level):
    for level in range(max_level + 1):
        w, h = (w + 1) // 2, (h + 1) // 2
        if w <= win[0] or h <= win[1]:
            return level
    return max_level


def ping_pong(n_ring, count):
    """0,1,..,n-1,n-2,..,1,0,1,.. : consecutive frames always differ by one motion step."""
    out, i, d = [], 0, 1
    for _ in range(count):
        out.append(i)
        if n_ring > 1:
            if i + d < 0 or i + d >= n_ring:
                d = -d
            i += d
    return out


def cpu_baseline(cfg, seconds_budget=25.0):
    """The CPU oracle (oracle/, kind "port") timed on this host over the same loop body: per pair one
    forward+backward LK of the live features (two pyramid builds + Scharr each, as OpenCV does) and every
    TRACK_LEN-th pair a detection.  Bounded sample; all host cores via OpenMP."""
    import oracle
    from iceberg_tracking_code_amd import synth
    oracle.build()
    cores = oracle.set_threads(0)
    w, h = cfg["w"], cfg["h"]
    # a horizontal band of the full frame keeps the sample bounded while every stage sees full-width rows
    band_h = min(h, 750)
    sh = synth.shifts(3, seed=1234)
    frames = [synth.frame(w, band_h, int(sx), int(sy), 1234) for sx, sy in sh]
    maxc = max(1, int(round(cfg["max_corners"] * band_h / h))) if cfg["max_corners"] > 0 else 0
    t0 = time.perf_counter()
    pairs, feats = 0, 0
    while True:
        pts = oracle.good_features(frames[0], maxc, DETECT["qualityLevel"], DETECT["minDistance"], None,
                                   DETECT["blockSize"])
        live = pts.reshape(-1, 2) if pts is not None else np.zeros((0, 2), np.float32)
        for k in range(TRACK_LEN):
            r = oracle.track_fb(frames[k], frames[k + 1], live, cfg["win"], cfg["max_level"], cfg["criteria"])
            feats += len(live)
            live = r["p1"][r["valid"].astype(bool)]
            pairs += 1
        if time.perf_counter() - t0 > seconds_budget * 0.5 or pairs >= 8:
            break
    dt = time.perf_counter() - t0
    scale = band_h / float(h)   # a band is band_h/h of a frame pair
    return dict(value=pairs * scale / dt, unit="frame-pairs/s", cores=cores, kind="port",
                tracked_features_per_sec=feats / dt,
                sample="%d pairs of a %dx%d band (%.0f%% of the %dx%d frame, features scaled alike) through "
                       "oracle/icelk_oracle.c: detect every %d pairs + forward/backward LK, OpenMP over %d threads"
                       % (pairs, w, band_h, 100 * scale, w, h, TRACK_LEN, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--ring", type=int, default=24, help="distinct frames resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="leave out the per-kernel HIP events (and with them the roofline object): shows what they cost")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="start a detection only when its frame is pushed (A/B of the cross-step overlap)")
    ap.add_argument("--source", default="hbm", choices=("hbm", "host"),
                    help="hbm: frames resident in HBM (the headline number); host: every frame crosses PCIe from "
                         "pinned host memory, uploads double-buffered against the tracker (BASELINE.json configs[2])")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: gloo / RCCL print banners to file descriptor 1, so everything but that
    # line is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # Control plane (barriers, max of the elapsed time) over gloo; the RCCL communicator is created only for the one
    # collective of the path, the final gather of the feature counts, AFTER the timed region: a handle already keeps
    # four HIP streams busy and a fifth queue on the device costs 25-35 %.
    dist = None
    if world > 1 or os.environ.get("ICELK_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from iceberg_tracking_code_amd import Context, SegmentTracker

    cfg = CONFIGS[args.config]
    w, h = cfg["w"], cfg["h"]
    K, W = args.steps, args.warmup
    ring = max(2, min(args.ring, K + W + 1))
    host = args.source == "host"
    if host:
        ring = min(ring, 8)
    max_pts = max(cfg["max_corners"], 1 << 14) if cfg["max_corners"] > 0 else 1 << 18
    ctx = Context(w, h, n_slots=5 if host else ring, max_pts=max_pts, device=local_rank)
    from iceberg_tracking_code_amd import synth
    shifts = synth.shifts(ring, seed=1234 + rank)
    pinned = []
    for i in range(ring):
        ctx.synth_frame(0 if host else i, w, h, int(shifts[i, 0]), int(shifts[i, 1]), 1234 + rank)
        if host:   # the same frames, parked in pinned host memory
            import ctypes
            img = np.ascontiguousarray(ctx.download_level(0, 0))
            ptr = ctx.host_alloc(w * h)
            ctypes.memmove(ptr, img.ctypes.data, w * h)
            pinned.append(ptr)
    ctx.sync()

    fp = dict(maxCorners=cfg["max_corners"], **DETECT)
    lk = dict(winSize=cfg["win"], maxLevel=cfg["max_level"], criteria=cfg["criteria"])
    tracker = SegmentTracker(w, h, TRACK_LEN, feature_params=fp, lk_params=lk, ctx=ctx,
                             lookahead=not args.no_lookahead)
    order = ping_pong(ring, K + W)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    def step(i):
        if not host:
            # the following frame is resident too: its detection (if it is a detection frame) may start now
            tracker.push_slot(order[i], wait=False, next_slot=order[i + 1] if i + 1 < W + K else None,
                              next2_slot=order[i + 2] if i + 2 < W + K else None,
                              next3_slot=order[i + 3] if i + 3 < W + K else None)
            return
        if i + 2 < W + K:   # frames i+1, i+2 are crossing PCIe while frame i is tracked
            tracker.prefetch_pinned(pinned[order[i + 2]], w)
        tracker.push_prefetched(wait=False)

    if host:
        tracker.prefetch_pinned(pinned[order[0]], w)
        tracker.prefetch_pinned(pinned[order[1]], w)
    for i in range(W):
        step(i)
    barrier()
    _, tracked0 = tracker.live()
    ctx.prof_reset()
    ctx.prof_enable(not args.no_kernel_timing)
    barrier()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    barrier()
    t1 = time.perf_counter()
    ctx.prof_enable(False)
    n_live, tracked1 = tracker.live()
    prof = ctx.prof_table()

    # the same kernels once more, each ALONE on the device (outside the timed region): inside the pipeline their
    # HIP-event durations include waiting for wave slots beside the tracker launch
    alone = {}
    if rank == 0 and not args.no_kernel_timing and not host:   # the resident ring provides the frames
        bs = DETECT["blockSize"]
        for _ in range(2):   # first pass warms up
            ctx.sync()
            ctx.prof_reset()
            ctx.prof_enable(True)
            for rep in range(5):      # back to back, one kind at a time
                s1 = order[(rep + 1) % len(order)]
                ctx.drop_pyramid(s1)
                ctx.build_pyramid(s1, cfg["win"], cfg["max_level"])
            ctx.sync()
            for rep in range(5):
                ctx.seg_detect_prepare(order[(rep + 1) % len(order)], False, bs)
            ctx.sync()
            for rep in range(5):
                s0, s1 = order[rep % len(order)], order[(rep + 1) % len(order)]
                tracker.ctx.seg_track(s0, s1, cfg["win"], cfg["max_level"], cfg["criteria"], 1e-4, 1.0, wait=False)
            ctx.sync()
            ctx.prof_enable(False)
            alone = ctx.prof_table()
    elapsed = t1 - t0
    tracked = tracked1 - tracked0

    # max over ranks of the elapsed time; the one collective of the path: RCCL all-gather of the per-rank
    # feature counts (sharding.gather_counts, also exercised under gloo in tests/test_host_logic.py)
    tracked_all = [tracked]
    if dist is not None:
        from iceberg_tracking_code_amd import sharding
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gather_backend = "rccl"
        try:
            rccl = dist.new_group(backend="nccl")     # RCCL over xGMI
            tracked_all = [int(v) for v in sharding.gather_counts([tracked], dist, device="cuda", group=rccl)]
        except Exception as exc:                      # the line is still worth printing: same counts over gloo
            sys.stderr.write("RCCL gather failed (%s); using gloo\n" % exc)
            gather_backend = "gloo (RCCL failed)"
            tracked_all = [int(v) for v in sharding.gather_counts([tracked], dist)]

    if rank == 0:
        top = top_level_of(w, h, cfg["win"], cfg["max_level"])
        pairs_per_s = world * K / elapsed
        feats_per_s = sum(tracked_all) / elapsed
        out = {
            "metric": "frame_pairs_per_sec", "value": pairs_per_s, "unit": "frame-pairs/s",
            "tracked_features_per_sec": feats_per_s,
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32/f32",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "width": w, "height": h, "max_corners": cfg["max_corners"],
                       "win": list(cfg["win"]), "maxLevel": cfg["max_level"], "pyramid_images": top + 1,
                       "criteria": list(cfg["criteria"]), "track_len": TRACK_LEN, "detector": DETECT,
                       "frames_resident": ring,
                       "source": "pinned host memory, hipMemcpyAsync double-buffered" if host else "HBM-resident",
                       "sharding": "independent segments per rank, no data-path collective",
                       "count_gather": gather_backend if dist is not None else None},
        }
        kern = {}
        lkp = prof.get("lk_fb")
        if lkp:
            n_avg = tracked / max(lkp["launches"], 1)
            alg = 2.0 * lk_algorithmic_bytes(w, h, cfg["win"], top, n_avg)   # forward + backward
            ach = alg / (lkp["avg_us"] * 1e-6) / 1e9
            out["roofline"] = {"kernel": "k_lk<fb> (fused forward+backward pyramidal LK, all levels)",
                               "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "algorithmic_bytes_per_launch": alg, "avg_launch_us": lkp["avg_us"],
                               "features_per_launch": n_avg,
                               "note": "LK is LDS/VALU-bound (SURVEY.md 8d); HBM fraction reported for completeness"}
        pd = prof.get("pyrdown")
        if pd:
            alg = pyramid_algorithmic_bytes(w, h, top)
            per_frame_us = pd["total_ms"] * 1e3 / K
            kern["pyramid"] = {"bound": "hbm", "algorithmic_bytes_per_frame": alg, "us_per_frame": per_frame_us,
                               "achieved_GBps": alg / (per_frame_us * 1e-6) / 1e9,
                               "frac": alg / (per_frame_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
            if "pyrdown" in alone:
                a_us = alone["pyrdown"]["total_ms"] * 1e3 / 5.0
                kern["pyramid"].update(alone_us_per_frame=a_us, alone_GBps=alg / (a_us * 1e-6) / 1e9,
                                       alone_frac=alg / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        eg = prof.get("corner_candidates")
        if eg:
            alg = 1.0 * w * h   # 1 B/px in; the eigenvalue map is never materialised (k_corners.hip)
            kern["corner_candidates"] = {"bound": "hbm", "algorithmic_bytes_per_launch": alg, "avg_launch_us": eg["avg_us"],
                               "achieved_GBps": alg / (eg["avg_us"] * 1e-6) / 1e9,
                               "frac": alg / (eg["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "note": "f64 box sums: issue-bound, not HBM-bound (DESIGN.md 4.2)"}
            if "corner_candidates" in alone:
                a_us = alone["corner_candidates"]["avg_us"]
                kern["corner_candidates"].update(alone_us=a_us, alone_GBps=alg / (a_us * 1e-6) / 1e9,
                                                 alone_frac=alg / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        if lkp and "lk_fb" in alone:
            kern["lk_fb"] = {"bound": "valu issue", "avg_launch_us": lkp["avg_us"], "alone_us": alone["lk_fb"]["avg_us"]}
        out["kernels"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2)} for k, v in prof.items()}
        out["kernel_rooflines"] = kern
        traffic_file = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if "roofline" in out and os.path.exists(traffic_file):
            try:
                tj = json.load(open(traffic_file))
                out["roofline"]["traffic"] = tj.get("lk_fb_bytes_per_launch")
                out["roofline"]["traffic_source"] = tj.get("source")
                if tj.get("lk_fb_valu_insts_per_launch") and tj.get("valu_issue_peak_per_s"):
                    # the bound this kernel actually runs into: VALU issue (PMC SQ_INSTS_VALU per launch, same file)
                    rate = tj["lk_fb_valu_insts_per_launch"] / (out["roofline"]["avg_launch_us"] * 1e-6)
                    out["roofline"]["valu_issue"] = {"wave_instructions_per_launch": tj["lk_fb_valu_insts_per_launch"],
                                                     "achieved_per_s": rate, "peak_per_s": tj["valu_issue_peak_per_s"],
                                                     "frac": rate / tj["valu_issue_peak_per_s"],
                                                     "peak_source": tj.get("valu_issue_peak_source")}
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for ptr in pinned:
        ctx.host_free(ptr)
    ctx.close()


if __name__ == "__main__":
    main()
