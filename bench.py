#!/usr/bin/env python3
"""bench.py -- frame-pairs/s and tracked-features/s of the sparse LK tracking loop on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 10               # BASELINE.json configs[1] (C2), the headline
    python bench.py --config c3                                    # configs[2]: 64 pairs streamed from pinned host memory
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # configs[3] (C4) shape: sharded sequence + RCCL gather

A "step" is one pass of the reference's frame loop body (s1_lucaskanade_tracking.py:307-450) over one new frame that is
already resident in HBM: Gaussian pyramid of the new frame, fused forward + backward pyramidal LK of all live features
against the previous frame, forward-backward filter and track-table append, and -- every `track_len` (= 2, s1:128)
frames -- Shi-Tomasi detection of new features (the reference's detector parameters s1:240-243; maxCorners capped at
10 000 as BASELINE.json configs[1] asks).  Nothing is skipped or cached inside the timed region; no host transfer of
images is in it either (`--config c3` reports the PCIe-inclusive rate beside it, never as `value`).

Multi-GPU (world > 1): the ranks shard ONE procedural sequence of world * K + 1 frames by segments
(sharding.frame_block; a segment = track_len + 1 frames from a detection frame, s1:362,440), no data-path collective
(weak scaling); after the timed region the per-segment track counts AND the padded track tables (the `tracks` arrays of
s1:394-395) are all-gathered over RCCL.  Barriers and the max over ranks of the elapsed time go over gloo.  rank 0
prints ONE JSON line.  An RCCL failure is reported (`count_gather_ok: false`) and makes the run exit non-zero.
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2_NAME = ("4000x3000 gray, 10k Shi-Tomasi features, winSize 21x21, maxLevel 3 (4 pyramid images), criteria (30, 0.01), "
           "track_len 2")
CONFIGS = {
    # BASELINE.json configs[1]: single MI355X, 4000x3000 gray pair, 10k Shi-Tomasi features, 21x21, 3 levels
    "c2": dict(w=4000, h=3000, max_corners=10000, win=(21, 21), max_level=3, criteria=(3, 30, 0.01), name="C2: " + C2_NAME),
    # configs[2]: batch of 64 consecutive 4000x3000 pairs streamed from host (hipMemcpyAsync double buffer)
    "c3": dict(w=4000, h=3000, max_corners=10000, win=(21, 21), max_level=3, criteria=(3, 30, 0.01),
               name="C3: 65 consecutive frames (64 pairs) of " + C2_NAME),
    # configs[3]: sequence sharded by segments over the ranks, RCCL gather of the track tables
    "c4": dict(w=4000, h=3000, max_corners=10000, win=(21, 21), max_level=3, criteria=(3, 30, 0.01),
               name="C4 (scaled to world x steps + 1 frames): procedural sequence of " + C2_NAME),
    # configs[4]
    "c5": dict(w=5760, h=3840, max_corners=50000, win=(31, 31), max_level=5, criteria=(3, 30, 0.01),
               name="C5: 5760x3840 gray, 50k features, winSize 31x31, maxLevel 5, track_len 2"),
    # the reference's own literals (s1:240-248) on its typical frame size (create_calibration_file.py:18)
    "ref": dict(w=3456, h=2304, max_corners=0, win=(35, 35), max_level=4, criteria=(3, 25, 0.03),
                name="REF: 3456x2304 gray, uncapped features, winSize 35x35, maxLevel 4, criteria (25, 0.03)"),
}
DETECT = dict(qualityLevel=0.007, minDistance=10, blockSize=10)   # s1:241-243
TRACK_LEN = 2                                                       # s1:128
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SIMDS, CLOCK_HZ = 1024, 2.4e9


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


def motion_tables(n, seed, motion):
    """(shifts, affines) of frames 0..n-1: a seeded random walk of sub-pixel shifts of up to 3 px per frame, and -- for
    motion == "shear" -- of affine coefficients within +-0.5 %, so that the displacement differs across
    the frame and the coarse pyramid levels have work to do."""
    from iceberg_tracking_code_amd import synth
    sh = synth.shifts(n, seed=seed)
    af = synth.affines(n, seed=seed) if motion == "shear" else np.zeros((n, 4), np.int64)
    return sh, af


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def usable_cores():
    """Cores this process may really use: the affinity mask, cut by the cgroup CPU quota (a GPU box hands a container a
    share of the host: os.cpu_count() still reports the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def probe_cv2():
    """probe, don't assume."""
    try:
        import cv2
        return dict(present=True, version=cv2.__version__, threads=int(cv2.getNumThreads()))
    except Exception as exc:   # ImportError here and on the GPU box
        return dict(present=False, error="%s: %s" % (type(exc).__name__, exc))


def _oracle_unit(oracle, frames, maxc, cfg):
    """One detection + TRACK_LEN forward/backward pairs of the reference loop body on the CPU oracle."""
    pts = oracle.good_features(frames[0], maxc, DETECT["qualityLevel"], DETECT["minDistance"], None, DETECT["blockSize"])
    live = pts.reshape(-1, 2) if pts is not None else np.zeros((0, 2), np.float32)
    feats = 0
    for k in range(TRACK_LEN):
        r = oracle.track_fb(frames[k], frames[k + 1], live, cfg["win"], cfg["max_level"], cfg["criteria"])
        feats += len(live)
        live = r["p1"][r["valid"].astype(bool)]
    return feats


def _cv2_unit(cv2, frames, maxc, cfg):
    """The reference's own call sequence (s1:323,326,329-333,437) on cv2, same frames."""
    fp = dict(maxCorners=maxc if maxc > 0 else 50000000, qualityLevel=DETECT["qualityLevel"],
              minDistance=DETECT["minDistance"], blockSize=DETECT["blockSize"])
    lk = dict(winSize=cfg["win"], maxLevel=cfg["max_level"], criteria=cfg["criteria"])
    p = cv2.goodFeaturesToTrack(frames[0], mask=None, **fp)
    feats = 0
    for k in range(TRACK_LEN):
        if p is None or len(p) == 0:
            break
        p1, _, _ = cv2.calcOpticalFlowPyrLK(frames[k], frames[k + 1], p, None, **lk)
        p0r, _, _ = cv2.calcOpticalFlowPyrLK(frames[k + 1], frames[k], p1, None, **lk)
        d = abs(p - p0r).reshape(-1, 2)
        feats += len(p)
        p = p1[np.hypot(d[:, 0], d[:, 1]) < 1]
    return feats


def cpu_baseline(cfg, motion):
    """the loop body timed on THIS host's cores in the same run -- 1 warm-up + median of 5, at all
    cores on the full frame and at 1 thread on a bounded sample (a full-width band of the frame, features scaled alike),
    the repo's own C restatement (oracle/, kind "port") always and the reference's cv2 call sequence too where cv2 imports."""
    import oracle
    from iceberg_tracking_code_amd import synth
    oracle.build()
    w, h = cfg["w"], cfg["h"]
    sh, af = motion_tables(TRACK_LEN + 1, 1234, motion)

    def band(rows):
        fr = [synth.frame(w, rows, int(sh[i, 0]), int(sh[i, 1]), 1234, affine=af[i]) for i in range(TRACK_LEN + 1)]
        mc = max(1, int(round(cfg["max_corners"] * rows / h))) if cfg["max_corners"] > 0 else 0
        return fr, mc

    def timed(fn, frames, mc, reps):
        fn(frames, mc)   # warm-up
        ts, feats = [], 0
        for _ in range(reps):
            t0 = time.perf_counter()
            feats = fn(frames, mc)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), feats

    out = dict(unit="frame-pairs/s", kind="port", cpu_model=cpu_model(), os_cpu_count=os.cpu_count(),
               usable_cores=usable_cores(), cv2=probe_cv2(),
               protocol="1 warm-up + median of 5 repeats of [1 detection + %d forward/backward pairs]" % TRACK_LEN)
    all_cores = oracle.set_threads(usable_cores())
    rows_all = h          # the all-cores leg runs the FULL frame (0.5-2 s per unit at 16 threads); the band is for 1 thread only
    fr, mc = band(rows_all)
    t_all, feats = timed(lambda f, m: _oracle_unit(oracle, f, m, cfg), fr, mc, 5)
    scale = rows_all / float(h)
    out.update(value=TRACK_LEN * scale / t_all, cores=all_cores, tracked_features_per_sec=feats / t_all,
               sample="the full %dx%d frame (%d features): 1 detection + %d forward/backward pairs through oracle/icelk_oracle.c, "
                      "OpenMP over %d threads" % (w, rows_all, mc, TRACK_LEN, all_cores))
    rows_1 = min(h, 250)
    fr1, mc1 = band(rows_1)
    oracle.set_threads(1)
    t_1, feats1 = timed(lambda f, m: _oracle_unit(oracle, f, m, cfg), fr1, mc1, 5)
    oracle.set_threads(all_cores)
    out["one_thread"] = dict(value=TRACK_LEN * (rows_1 / float(h)) / t_1, cores=1, tracked_features_per_sec=feats1 / t_1,
                             sample="a %dx%d band (%.1f%% of the frame, %d features)" % (w, rows_1, 100.0 * rows_1 / h, mc1))
    if out["cv2"]["present"]:   # never silently substituted: reported beside the port, kind "reference"
        import cv2
        res = {}
        for nthr, (frs, mcs, rows) in ((all_cores, (fr, mc, rows_all)), (1, (fr1, mc1, rows_1))):
            cv2.setNumThreads(nthr)
            t, fe = timed(lambda f, m: _cv2_unit(cv2, f, m, cfg), frs, mcs, 5)
            res["threads_%d" % nthr] = dict(value=TRACK_LEN * (rows / float(h)) / t, tracked_features_per_sec=fe / t)
        out["cv2_reference"] = res
    else:
        out["note"] = "OpenCV unavailable on host -- CPU baseline is the repo's own restatement"
    return out


def numa_placement(device_index):
    """NUMA node of the device (sysfs, by PCI bus id) and of the CPU this process runs on -- pinned buffers are placed by
    the allocating thread's policy; a remote node costs the copy engine bandwidth.  None where sysfs does not say."""
    out = {"cpu": None, "cpu_node": None, "gpu_node": None}
    try:
        out["cpu"] = cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else None
        import glob
        for nd in glob.glob("/sys/devices/system/node/node[0-9]*"):
            cpus = set()
            for part in open(nd + "/cpulist").read().strip().split(","):
                if part:
                    a, _, b = part.partition("-")
                    cpus.update(range(int(a), int(b or a) + 1))
            if cpu in cpus:
                out["cpu_node"] = int(nd.rsplit("node", 1)[1])
        import torch
        bus = torch.cuda.get_device_properties(device_index).pci_bus_id if hasattr(torch.cuda.get_device_properties(device_index), "pci_bus_id") else None
        dom = getattr(torch.cuda.get_device_properties(device_index), "pci_domain_id", 0)
        dev = getattr(torch.cuda.get_device_properties(device_index), "pci_device_id", 0)
        if bus is not None:
            path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (dom, bus, dev)
            out["gpu_node"] = int(open(path).read().strip())
    except Exception as e:       # diagnostics only
        out["error"] = repr(e)
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU), BEFORE anything in
    this process has touched the GPU -- no exec of a process that has initialised HIP, the children are plain child
    processes --, relay rank 0's JSON line and exit non-zero if any rank does.  Equivalent to
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...`."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if env.get("ICELK_BENCH_SHARED_DEVICE"):
            env["LOCAL_RANK"] = "0"      # rehearsal on one GPU (tools/two_rank.sh): every rank on device 0, RCCL leg skipped
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0]
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.buffer.write(out0)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks exited non-zero: %s\n" % bad)
        return bad[0][1] if bad[0][1] > 0 else 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c2 on one GPU, c4 (sharded sequence + RCCL gather) on several")
    ap.add_argument("--ring", type=int, default=64, help="distinct frames resident in HBM (c2 / c5 / ref)")
    ap.add_argument("--motion", default="shear", choices=("shear", "translate"),
                    help="frame-to-frame motion of the synthetic sequence: sub-pixel translation plus a slowly varying "
                         "affine deformation of up to 0.5 %% (default), or translation only (the easiest case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="leave out the per-kernel HIP events (and with them the roofline object): shows what they cost")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="HIP-event timing of every kernel inside the timed region (default: the tracker launches only)")
    ap.add_argument("--pcie-depth", type=int, default=6, help="c3: uploads in flight ahead of the tracker in the PCIe-inclusive run")
    ap.add_argument("--pcie-spare-slots", type=int, default=6, help="c3: slots beyond the uploads in flight (previous + current + spares)")
    ap.add_argument("--no-pair-launch", action="store_true",
                    help="A/B: the last pair of a segment is launched on its own instead of with the first pair of the next")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="start a detection only when its frame is pushed (A/B of the cross-step overlap)")
    ap.add_argument("--lk-sums", type=int, default=0, choices=(0, 1, 2),
                    help="icelk_set_variant('lk_sums'): 0 = exact sums (default), 1 / 2 = the float-lane order of OpenCV 3.x's SSE2 / "
                         "4.x's CV_SIMD128 block (DESIGN.md section 2); the rate under a variant is reported, never the headline")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="keep the device busy with the same loop for this long BEFORE the W warm-up steps (untimed): the device's "
                         "clocks take tens of milliseconds of load to settle, and a 20-step window is 3.5 ms long "
                         "(profiles/r04_short_run_sweep.txt).  0: none")
    ap.add_argument("--no-archive", action="store_true",
                    help="A/B: finished segments are not compacted into the device archive inside the timed region "
                         "(the loop's output, s1:394-395, is then left out of it)")
    args = ap.parse_args()
    echo_dir = os.environ.get("ICELK_BENCH_SPAWN_ECHO")     # tests/test_host_logic.py: what a spawned rank is handed
    if echo_dir and "WORLD_SIZE" in os.environ:
        info = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        info.update(gpus=args.gpus, steps=args.steps, warmup=args.warmup)
        with open(os.path.join(echo_dir, "rank_%s.json" % info["RANK"]), "w") as f:
            json.dump(info, f)
        print(json.dumps(info))
        sys.exit(int(os.environ.get("ICELK_BENCH_SPAWN_ECHO_FAIL", "0")) if info["RANK"] == "1" else 0)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    # stdout carries exactly one JSON line: gloo / RCCL print banners to file descriptor 1, so everything but that
    # line is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch one process per GPU (torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # Control plane (barriers, max of the elapsed time) over gloo; the RCCL communicator is created only for the
    # collectives of the path, the final gathers, AFTER the timed region: a handle already keeps four HIP streams busy
    # and a fifth queue on the device costs 25-35 %.
    dist = None
    if world > 1 or os.environ.get("ICELK_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from iceberg_tracking_code_amd import Context, SegmentTracker, sharding

    cname = args.config or ("c2" if world == 1 else "c4")
    cfg = CONFIGS[cname]
    w, h = cfg["w"], cfg["h"]
    W = args.warmup
    K = args.steps if args.steps is not None else (64 if cname == "c3" else 200)
    if cname == "c3":
        K = 64   # BASELINE.json configs[2]: 64 consecutive pairs
    linear = cname in ("c3", "c4")            # a sequence of distinct frames instead of a ping-pong ring
    steps_note = None
    if linear and K % TRACK_LEN:
        # a rank's block of the sequence starts and ends at a detection frame (sharding.frame_block): the timed steps are
        # rounded down to whole segments and the line says so -- a scaling run must not die on an odd --steps
        K_req, K = K, max(TRACK_LEN, K - K % TRACK_LEN)
        steps_note = "--steps %d rounded to %d: %s shards by segments of track_len = %d frames" % (K_req, K, cname, TRACK_LEN)
        sys.stderr.write("bench.py: " + steps_note + "\n")
    ring = (W + K + 1) if linear else max(2, min(args.ring, K + W + 1))
    max_pts = max(cfg["max_corners"], 1 << 14) if cfg["max_corners"] > 0 else 1 << 18
    seed = 1234
    # frame g of the GLOBAL sequence (g = 0 is the first timed frame of rank 0; the W warm-up frames precede a rank's
    # block); ring configs use frames 0..ring-1 of a per-rank sequence
    if linear:
        # the rank's frames of the one global sequence: its block of segments plus the closing frame (sharding.frame_block)
        f0, f1 = sharding.frame_block(world * K + 1, TRACK_LEN, rank, world)
        assert f1 - f0 == K + 1, (f0, f1, K)
        sh_all, af_all = motion_tables(world * K + 1 + W, seed, args.motion)
        sh, af = sh_all[f0:f0 + ring], af_all[f0:f0 + ring]
    else:
        sh, af = motion_tables(ring, seed + rank, args.motion)
    ctx = Context(w, h, n_slots=ring, max_pts=max_pts, device=local_rank)
    if args.lk_sums:
        ctx.set_variant("lk_sums", args.lk_sums)
    probe_info = ctx.stream_probe_info()
    for i in range(ring):
        ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), seed if linear else seed + rank, affine=af[i])
    ctx.sync()

    fp = dict(maxCorners=cfg["max_corners"], **DETECT)
    lk = dict(winSize=cfg["win"], maxLevel=cfg["max_level"], criteria=cfg["criteria"])

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    def run_resident(tracker, order, first, count):
        """`count` steps over frames that sit in HBM; the slots of the next six frames are known to the tracker."""
        n = len(order)
        for i in range(first, first + count):
            tracker.push_slot(order[i], False, *[order[i + k] if i + k < n else None for k in range(1, 7)])

    # ---- the timed region -------------------------------------------------------------------------------------------------
    # ring configs: the visiting order runs on beyond the timed steps, so that the last timed steps look ahead (corner
    # candidates, min-distance stages, pyramids and joint launches for the frames that follow) exactly as the warm-up looked
    # ahead for the first timed steps -- the timed region is a window of the steady state: K pairs go out inside it
    settle_max = int(args.settle_ms * 1e3 / 100.0) + 64 if args.settle_ms > 0 else 0   # steps, were a step to last only 100 us
    order = list(range(ring)) if linear else ping_pong(ring, settle_max + K + W + SegmentTracker.MAX_AHEAD)
    archive = None
    settle = dict(ms_asked=args.settle_ms, ms=0.0, steps=0,
                  note="untimed: the same loop before the W warm-up steps, so that the timed window sees settled clocks")

    def ring_archive(tracker):
        """The loop's output (s1:394-395: one `tracks` / `trackquality` pair per finished segment): every finished segment is
        compacted, in track order, into a ring of archive entries on the device (icelk_seg_archive; what C4 gathers over
        RCCL at the end) -- inside the timed region."""
        rows = max_pts if cfg["max_corners"] <= 0 else cfg["max_corners"]
        n_ring = 8
        a = dict(tracks=torch.zeros((n_ring, rows, TRACK_LEN + 1, 2), dtype=torch.float32, device="cuda"),
                 quality=torch.zeros((n_ring, rows, TRACK_LEN), dtype=torch.float32, device="cuda"),
                 counts=torch.zeros(n_ring, dtype=torch.int32, device="cuda"), n=0, rows=rows, ring=n_ring)

        def on_close_ring(first_frame, closed):
            s = a["n"] % a["ring"]
            ctx.seg_archive(a["tracks"][s].data_ptr(), a["quality"][s].data_ptr(), a["counts"][s:s + 1].data_ptr(), a["rows"],
                            closed=closed)
            a["n"] += 1
        tracker.on_close = on_close_ring
        return a
    if linear:
        # warm-up on the W frames before the rank's block with a tracker of its own that starts nothing ahead of time:
        # the timed tracker begins with a clean handle (no detection in flight, no staged segment)
        ts0 = time.perf_counter()
        while args.settle_ms > 0 and W > 0 and (time.perf_counter() - ts0) * 1e3 < args.settle_ms:
            pre = SegmentTracker(w, h, TRACK_LEN, feature_params=fp, lk_params=lk, ctx=ctx, lookahead=False)
            run_resident(pre, order[:W], 0, W)
            ctx.sync()
            pre.abort()
            settle["steps"] += W
        settle["ms"] = (time.perf_counter() - ts0) * 1e3
        warm = SegmentTracker(w, h, TRACK_LEN, feature_params=fp, lk_params=lk, ctx=ctx, lookahead=False)
        run_resident(warm, order[:W], 0, W)
        tracker = SegmentTracker(w, h, TRACK_LEN, feature_params=fp, lk_params=lk, ctx=ctx, lookahead=not args.no_lookahead, pair_launch=not args.no_pair_launch)
        if cname == "c4":
            # the rank's finished segments stay on the device: (segments, rows, vertices, 2) float32 + counts, gathered
            # over RCCL after the timed region (icelk_seg_archive: no host round trip per segment)
            n_seg = sharding.segment_count(K + 1, TRACK_LEN)
            rows = max_pts if cfg["max_corners"] <= 0 else cfg["max_corners"]
            archive = dict(tracks=torch.zeros((max(n_seg, 1), rows, TRACK_LEN + 1, 2), dtype=torch.float32, device="cuda"),
                           counts=torch.zeros(max(n_seg, 1), dtype=torch.int32, device="cuda"), n=0, rows=rows)

            def on_close(first_frame, closed):
                s = archive["n"]
                if s < archive["tracks"].shape[0]:
                    ctx.seg_archive(archive["tracks"][s].data_ptr(), 0, archive["counts"][s:s + 1].data_ptr(), archive["rows"],
                                    closed=closed)
                    archive["n"] = s + 1
            tracker.on_close = on_close
        elif not args.no_archive:
            archive = ring_archive(tracker)
        timed_order = order[W:]
        t_first, pushes = 0, K + 1    # the first push of a fresh tracker only detects: K + 1 frames = K frame pairs
    else:
        tracker = SegmentTracker(w, h, TRACK_LEN, feature_params=fp, lk_params=lk, ctx=ctx, lookahead=not args.no_lookahead, pair_launch=not args.no_pair_launch)
        if not args.no_archive:
            archive = ring_archive(tracker)
        ts0, S = time.perf_counter(), 0
        while args.settle_ms > 0 and S + 32 <= settle_max and (time.perf_counter() - ts0) * 1e3 < args.settle_ms:
            run_resident(tracker, order, S, 32)
            ctx.sync()
            S += 32
        settle["steps"], settle["ms"] = S, (time.perf_counter() - ts0) * 1e3
        run_resident(tracker, order, S, W)
        timed_order = order
        t_first, pushes = S + W, K    # steady state: every push tracks one pair
    barrier()
    _, tracked0 = tracker.live() if tracker.active else (0, 0)
    if linear:
        tracked0 = ctx.seg_live()[1] if warm.active else 0
    ctx.prof_reset()
    # the tracker launches are timed (HIP events on their stream); timing every kernel of the pipeline costs two event
    # records per kernel on the detector's streams and 7 % of the throughput: --time-all-kernels
    ctx.prof_enable(0 if args.no_kernel_timing else (1 if args.time_all_kernels else 2))
    barrier()
    archived0 = archive["n"] if archive else 0
    pairs0 = tracker.pairs_launched
    t0 = time.perf_counter()
    run_resident(tracker, timed_order, t_first, pushes)
    tracker.flush()    # a last pair held back for a joint launch goes out (and is archived) inside the timed region
    barrier()
    t1 = time.perf_counter()
    ctx.prof_enable(False)
    prof = ctx.prof_table()
    archived = (archive["n"] - archived0) if archive else 0
    archiving = archive is not None
    pairs_by_tracker = tracker.pairs_launched - pairs0
    tail_stats, tmpl_info = ctx.seg_tail_stats(), ctx.seg_template_info()
    consumed = tracker.abort()      # what was started ahead for frames beyond the timed steps is abandoned (it has run)
    n_live, tracked1 = tracker.live()
    elapsed = t1 - t0
    tracked = tracked1 - tracked0
    # frame pairs whose tracker launch lies inside the timed region (a joint launch carries two)
    # counted by the tracker itself (the same quantity with and without --no-kernel-timing); the HIP-event table of the
    # tracker launches must agree where it exists
    pairs_launched = pairs_by_tracker
    pairs_by_events = None
    if not args.no_kernel_timing:
        pairs_by_events = (prof.get("lk_fb", {}).get("launches", 0) + 2 * prof.get("lk_fb_pair", {}).get("launches", 0))
    pairs_timed = pairs_launched
    # segments that closed inside the timed region: detection frames among the timed steps (the first frame of a fresh
    # tracker only starts a segment)
    first_counter = 0 if linear else t_first
    segments_expected = len([c for c in range(first_counter, first_counter + pushes) if c % TRACK_LEN == 0 and c > 0])

    # ---- the same kernels once more, each ALONE on the device (outside the timed region): inside the pipeline their
    # HIP-event durations include waiting for wave slots beside the tracker launch ------------------------------------------
    alone, iters, pcie = {}, None, None
    if rank == 0 and not args.no_kernel_timing:
        bs = DETECT["blockSize"]
        sl = list(range(min(ring, 6)))
        for _ in range(2):   # first pass warms up
            ctx.sync()
            ctx.prof_reset()
            ctx.prof_enable(True)
            for rep in range(5):      # back to back, one kind at a time
                ctx.drop_pyramid(sl[(rep + 1) % len(sl)])
                ctx.build_pyramid(sl[(rep + 1) % len(sl)], cfg["win"], cfg["max_level"])
            ctx.sync()
            for rep in range(5):
                ctx.seg_detect_prepare(sl[(rep + 1) % len(sl)], False, bs)
            ctx.sync()
            for rep in range(5):
                ctx.seg_track(sl[rep % len(sl)], sl[(rep + 1) % len(sl)], cfg["win"], cfg["max_level"], cfg["criteria"], 1e-4,
                              1.0, wait=False)
            ctx.sync()
            ctx.prof_enable(False)
            alone = ctx.prof_table()
        # iterations per feature (SURVEY.md 8d "so this is checkable"): one fresh segment on the bench's own frames
        ctx.prof_enable(True)
        nd = ctx.seg_detect(sl[0], cfg["max_corners"], DETECT["qualityLevel"], DETECT["minDistance"], False, bs)
        ctx.seg_track(sl[0], sl[1], cfg["win"], cfg["max_level"], cfg["criteria"], 1e-4, 1.0)
        itf, itb = ctx.prof_iterations()
        ctx.prof_enable(False)
        if len(itf):
            top = top_level_of(w, h, cfg["win"], cfg["max_level"])
            tot = itf + itb
            iters = dict(features=int(nd), levels=top + 1, mean_per_level_forward=float(itf.mean()) / (top + 1),
                         mean_per_level_backward=float(itb.mean()) / (top + 1),
                         histogram_forward_plus_backward={"bin_width": 4, "counts": np.bincount(tot // 4).tolist()},
                         max=int(tot.max()), criteria_max_count=cfg["criteria"][1])

        # gray conversion alone (K1; HBM-bound stream): a 3-channel device image of the frame size
        rgb = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.prof_reset()
        ctx.prof_enable(True)
        for rep in range(6):
            ctx.cvt_bgr_device(sl[-1], rgb.data_ptr(), w, h, 3 * w)   # overwrites that slot: nothing reads it afterwards
        ctx.sync()
        ctx.prof_enable(False)
        alone["bgr2gray"] = ctx.prof_table().get("bgr2gray")
        del rgb
    # ---- C3: the same 64 pairs once more, every frame crossing PCIe from pinned host memory ------------------------------
    if cname == "c3" and rank == 0:
        import ctypes
        pinned_all = []
        for i in range(ring):
            img = np.ascontiguousarray(ctx.download_level(i, 0))
            ptr = ctx.host_alloc(w * h)
            ctypes.memmove(ptr, img.ctypes.data, w * h)
            pinned_all.append(ptr)
        pinned = pinned_all[W:]
        ctx.close()   # one handle (four streams) on the device at a time
        ctx = None
        depth = args.pcie_depth   # uploads in flight ahead of the frame being tracked
        n_up_slots = depth + args.pcie_spare_slots   # previous, current, `depth` coming, and spares: the slot an upload
        # overwrites was last read several launches ago, so the copy never waits for the launch that is running
        hctx = Context(w, h, n_slots=n_up_slots, max_pts=max_pts, device=local_rank)
        hprobe = hctx.stream_probe_info()
        # the W warm-up frames through this handle too (same loop, same source), as the resident run had them
        hw = SegmentTracker(w, h, TRACK_LEN, feature_params=fp, lk_params=lk, ctx=hctx, lookahead=False)
        for i in range(W):
            hw.push_pinned(pinned_all[i], w, wait=False)
        hw.abort()
        ht = SegmentTracker(w, h, TRACK_LEN, feature_params=fp, lk_params=lk, ctx=hctx, lookahead=not args.no_lookahead,
                            pair_launch=not args.no_pair_launch)
        hctx.sync()
        torch.cuda.synchronize()
        _, htracked0 = hctx.seg_live() if hw.active else (0, 0)
        # what the link gives THIS process for THESE buffers, with nothing else on the device: 16 uploads into slots nothing
        # reads (the loop below starts by overwriting them); and where the buffers and the device sit (NUMA nodes)
        spare = [s_ for s_ in range(n_up_slots) if s_ not in (hw.cur, (hw.cur - 1) % n_up_slots)]
        u0 = time.perf_counter()
        for i in range(16):
            hctx.upload_gray_async(spare[i % len(spare)], pinned[i % len(pinned)], w, h, w)
        hctx.sync()
        raw_upload_us = 1e6 * (time.perf_counter() - u0) / 16
        numa = numa_placement(local_rank)
        h0 = time.perf_counter()
        for i in range(min(depth, len(pinned))):
            ht.prefetch_pinned(pinned[i], w)
        step_t = [h0]
        for i in range(len(pinned)):
            if i + depth < len(pinned):
                ht.prefetch_pinned(pinned[i + depth], w)
            ht.push_prefetched(wait=False)
            step_t.append(time.perf_counter())
        hctx.sync()
        torch.cuda.synchronize()
        h1 = time.perf_counter()
        step_us = np.diff(np.array(step_t + [h1])) * 1e6     # the host's time per frame; the last entry is the final wait
        _, htracked = ht.live()
        htracked -= htracked0
        # same frames, same loop: the survivors must agree with the resident run
        pcie = dict(value=(len(pinned) - 1) / (h1 - h0), unit="frame-pairs/s", pairs=len(pinned) - 1,
                    tracked_features_per_sec=htracked / (h1 - h0), live_tracks_equal_resident_run=bool(ht.live()[0] == n_live),
                    source="pinned host memory, hipMemcpyAsync, %d uploads in flight ahead of the tracker (%d slots)" % (depth, n_up_slots),
                    bytes_per_frame=w * h, stream_probe=hprobe, raw_upload_us=raw_upload_us,
                    raw_upload_GBps=w * h / raw_upload_us / 1e3, numa=numa,
                    host_us_per_frame=dict(median=float(np.median(step_us[:-1])), longest=float(step_us[:-1].max()),
                                           longest_at=int(step_us[:-1].argmax()), first8=[round(float(x), 1) for x in step_us[:8]],
                                           final_wait=float(step_us[-1])), note="a handle of its own, warmed up with the same W frames; includes the first frame's upload and the first "
                         "(blocking) detection of the 65-frame batch")
        for ptr in pinned_all:
            hctx.host_free(ptr)
        hctx.close()

    # ---- the collectives of the path: per-segment track counts and the padded track tables, over RCCL -------------------
    tracked_all, gather = [tracked], None
    if cname != "c4":
        archive_ring, archive = archive, None     # the gathers below are C4's: one archive entry per segment of the block
    else:
        archive_ring = archive
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        seg_counts = archive["counts"][:archive["n"]].cpu().numpy().tolist() if archive else [tracked]
        share = os.environ.get("ICELK_BENCH_SHARED_DEVICE")   # tools/two_rank.sh: both ranks on one GPU, RCCL cannot form
        ok, err, backend, rccl_ranks = True, "", "rccl", None
        g0 = time.perf_counter()
        res_counts, res_tables = None, None
        if share:
            ok, backend = True, "gloo (RCCL leg skipped: ranks share a device)"
        else:
            try:
                rccl = dist.new_group(backend="nccl")     # RCCL over xGMI
                rccl_ranks = int(dist.get_world_size(rccl))
                res_counts = sharding.gather_counts(seg_counts + [tracked], dist, device="cuda", group=rccl)
                if archive:
                    res_tables = sharding.gather_tables(archive["tracks"][:max(archive["n"], 1)], archive["counts"][:max(archive["n"], 1)],
                                                        dist, group=rccl)
                torch.cuda.synchronize()
            except Exception as exc:
                ok, err = False, "%s: %s" % (type(exc).__name__, exc)
                sys.stderr.write("rank %d: RCCL gather failed (%s)\n" % (rank, err))
        # every rank must take the same branch: agree on the outcome over the control group before touching it
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_ok = bool(flag.item())
        if share or not all_ok:
            if not share:
                backend = "gloo (RCCL failed)"
            res_counts = sharding.gather_counts(seg_counts + [tracked], dist)
            if archive:
                res_tables = sharding.gather_tables(archive["tracks"][:max(archive["n"], 1)].cpu(),
                                                    archive["counts"][:max(archive["n"], 1)].cpu(), dist)
        g1 = time.perf_counter()
        # each rank appended its feature total behind its segment counts
        per_rank = len(seg_counts) + 1
        rc = np.asarray(res_counts, np.int64).reshape(world, per_rank) if len(res_counts) == world * per_rank else None
        tracked_all = rc[:, -1].tolist() if rc is not None else [tracked]
        gather = dict(backend=backend, ok=bool(all_ok or share), seconds=g1 - g0, error=err or None, rccl_ranks=rccl_ranks,
                      segments_gathered=len(res_tables) if res_tables is not None else None,
                      segments=int(sum(len(seg_counts) for _ in range(world))) if archive else 0,
                      tracks_in_tables=int(sum(n for _, n in res_tables)) if res_tables is not None else None,
                      table_bytes_per_rank=int(archive["tracks"][:max(archive["n"], 1)].numel() * 4) if archive else 0)
        if res_tables is not None and rc is not None:
            gather["tables_match_counts"] = bool([n for _, n in res_tables] == rc[:, :-1].reshape(-1).tolist())

    exit_code = 0
    if rank == 0:
        top = top_level_of(w, h, cfg["win"], cfg["max_level"])
        # every rank runs the same schedule: world x the pairs rank 0 launched inside its timed region
        pairs_per_s = world * pairs_timed / elapsed
        feats_per_s = sum(tracked_all) / elapsed
        out = {
            "metric": "frame_pairs_per_sec", "value": pairs_per_s, "unit": "frame-pairs/s",
            "tracked_features_per_sec": feats_per_s,
            "n_gpus": world, "steps": K, "steps_note": steps_note, "warmup": W, "settle": settle, "ms_per_step": 1e3 * elapsed / max(pairs_timed, 1),
            "pairs_launched_in_timed_region": pairs_launched, "pairs_expected": K,
            "pairs_launched_by_hip_events": pairs_by_events,
            "pairs_mismatch": bool(pairs_launched != K or (pairs_by_events is not None and pairs_by_events != pairs_launched)),
            "segments_archived_in_timed_region": archived if archiving else None, "segments_expected": segments_expected,
            "archive": ("every finished segment compacted on the device inside the timed region (icelk_seg_archive: tracks + "
                        "trackquality of s1:394-395)" if archiving else "off (--no-archive)"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32/f32",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "width": w, "height": h, "max_corners": cfg["max_corners"],
                       "win": list(cfg["win"]), "maxLevel": cfg["max_level"], "pyramid_images": top + 1,
                       "criteria": list(cfg["criteria"]), "track_len": TRACK_LEN, "detector": DETECT,
                       "frames_resident": ring, "source": "HBM-resident", "lk_sums_variant": args.lk_sums,
                       "motion": "sub-pixel translation (<= 3 px/frame) + affine deformation (<= 0.5 %)" if args.motion == "shear"
                                 else "sub-pixel translation (<= 3 px/frame) only",
                       "sharding": "one sequence of %d frames, segment blocks per rank (sharding.frame_block), no data-path "
                                   "collective" % (world * K + 1) if linear and world > 1 else "single process",
                       "count_gather": gather["backend"] if gather else None},
            # which hardware queues the handle's side streams landed on: throughput depends on it, results do not
            "stream_probe": probe_info,
            # which tail staged the segments of this handle so far (k_tail.hip / the host's: icelk_seg_tail_stats), and the
            # template tables (icelk_seg_template_info: bytes of one of the two, rows, state)
            "detection_tails": {"device_driven": tail_stats[0], "host": tail_stats[1]},
            "template_tables": {"bytes_each": tmpl_info[0], "rows": tmpl_info[1], "state": tmpl_info[2]},
        }
        if gather:
            out["gather"] = gather
            out["count_gather_ok"] = gather["ok"]
            if not gather["ok"]:
                exit_code = 3
        if pcie:
            out["pcie_inclusive"] = pcie
        kern = {}
        # tracker launches of the timed region: single segment pairs and joint launches of two (icelk_seg_track_defer)
        lk_kinds = [prof[k] for k in ("lk_fb", "lk_fb_pair") if prof.get(k) and prof[k]["launches"]]
        lkp = None
        if lk_kinds:
            nl = sum(k["launches"] for k in lk_kinds)
            tot_ms = sum(k["total_ms"] for k in lk_kinds)
            lkp = {"launches": nl, "total_ms": tot_ms, "avg_us": tot_ms * 1e3 / nl,
                   "pairs_per_launch": (sum(k["launches"] for k in lk_kinds) + (prof.get("lk_fb_pair") or {"launches": 0})["launches"]) / nl}
        if lkp:
            # algorithmic bytes per forward+backward PAIR (per-level cap = 2 frames of that pair) x the pairs
            # one launch carries
            n_pair = tracked / max(pairs_timed, 1)
            alg_pair = 2.0 * lk_algorithmic_bytes(w, h, cfg["win"], top, n_pair)
            alg = alg_pair * lkp["pairs_per_launch"]
            n_avg = n_pair * lkp["pairs_per_launch"]
            ach = alg / (lkp["avg_us"] * 1e-6) / 1e9
            out["roofline"] = {"kernel": "k_lk_fast<fb> (fused forward+backward pyramidal LK, all levels)",
                               "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_pair": alg_pair,
                               "avg_launch_us": lkp["avg_us"],
                               "features_per_launch": n_avg, "frame_pairs_per_launch": lkp["pairs_per_launch"],
                               "note": "LK is VALU-issue-bound (SURVEY.md 8d); HBM fraction reported for completeness, the "
                                       "issue fraction is in valu_issue"}
        pd = prof.get("pyrdown")
        if pd or alone.get("pyrdown"):
            alg = pyramid_algorithmic_bytes(w, h, top)
            kern["pyramid"] = {"bound": "hbm", "algorithmic_bytes_per_frame": alg}
            if pd:
                per_frame_us = pd["total_ms"] * 1e3 / K
                kern["pyramid"].update(us_per_frame=per_frame_us, achieved_GBps=alg / (per_frame_us * 1e-6) / 1e9,
                                       frac=alg / (per_frame_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
            if alone.get("pyrdown"):
                a_us = alone["pyrdown"]["total_ms"] * 1e3 / 5.0
                kern["pyramid"].update(alone_us_per_frame=a_us, alone_GBps=alg / (a_us * 1e-6) / 1e9,
                                       alone_frac=alg / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                       alone_note="HIP events around each launch: 2-2.5 us more than the kernel itself lasts (rocprof_kernel_us)")
            # the kernel's own duration (rocprofv3 --kernel-trace --stats of tools/pyr_alone.py at 4000x3000, committed)
            stats = os.path.join(ROOT, "profiles", "r04_pyramid_alone_kernel_stats.csv")
            if cname in ("c2", "c3", "c4") and os.path.exists(stats):
                try:
                    import csv
                    row = [r for r in csv.DictReader(open(stats)) if "k_pyramid" in r["Name"]][0]
                    k_us = float(row["AverageNs"]) / 1e3
                    kern["pyramid"].update(rocprof_kernel_us=k_us, rocprof_frac=alg / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                           rocprof_source="profiles/r04_pyramid_alone_kernel_stats.csv (a committed profile, not this run)")
                except Exception:
                    pass
        if alone.get("bgr2gray"):
            alg = 4.0 * w * h
            a_us = alone["bgr2gray"]["avg_us"]
            kern["bgr2gray"] = {"bound": "hbm", "algorithmic_bytes_per_frame": alg, "alone_us": a_us,
                                "alone_GBps": alg / (a_us * 1e-6) / 1e9, "alone_frac": alg / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "alone_note": "HIP events around each launch: several microseconds more than the kernel itself lasts"}
            stats = os.path.join(ROOT, "profiles", "r04_gray_alone_kernel_stats.csv")
            if cname in ("c2", "c3", "c4") and os.path.exists(stats):
                try:
                    import csv
                    row = [r for r in csv.DictReader(open(stats)) if "k_bgr2gray" in r["Name"]][0]
                    k_us = float(row["AverageNs"]) / 1e3
                    kern["bgr2gray"].update(rocprof_kernel_us=k_us, rocprof_frac=alg / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                            rocprof_source="profiles/r04_gray_alone_kernel_stats.csv (tools/gray_alone.py; a committed profile, not this run)")
                except Exception:
                    pass
        eg = prof.get("corner_candidates")
        if eg or alone.get("corner_candidates"):
            alg = 1.0 * w * h   # 1 B/px in; the eigenvalue map is never materialised (k_corners.hip)
            kern["corner_candidates"] = {"bound": "hbm", "algorithmic_bytes_per_launch": alg,
                                         "note": "f64 box sums: issue-bound, not HBM-bound (DESIGN.md 4.2)"}
            if eg:
                kern["corner_candidates"].update(avg_launch_us=eg["avg_us"], achieved_GBps=alg / (eg["avg_us"] * 1e-6) / 1e9,
                                                 frac=alg / (eg["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
            if alone.get("corner_candidates"):
                a_us = alone["corner_candidates"]["avg_us"]
                kern["corner_candidates"].update(alone_us=a_us, alone_GBps=alg / (a_us * 1e-6) / 1e9,
                                                 alone_frac=alg / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        if lkp and alone.get("lk_fb"):
            kern["lk_fb"] = {"bound": "valu issue", "avg_launch_us": lkp["avg_us"], "alone_us": alone["lk_fb"]["avg_us"]}
        out["kernels"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2)} for k, v in prof.items()}
        out["kernel_rooflines"] = kern
        if iters:
            out["iterations_per_feature"] = iters
        pmc_file = os.path.join(ROOT, "profiles", "pmc_%s.json" % cname)
        if "roofline" in out and os.path.exists(pmc_file):
            # counters of THIS kernel from rocprofv3 --pmc runs of `python3 bench.py` (tools/collect_profiles.sh writes the
            # file; profiles/ holds the raw summaries it was made from)
            try:
                pj = json.load(open(pmc_file))
                # the counters are per launch of the profiled run (joint launches: two frame pairs each); the launches of
                # this run carry pairs_per_launch on average
                scale = lkp["pairs_per_launch"] / float(pj.get("frame_pairs_per_launch", 1))
                out["roofline"]["traffic"] = pj.get("lk_fb_bytes_per_launch") * scale if pj.get("lk_fb_bytes_per_launch") else None
                out["roofline"]["traffic_source"] = pj.get("source")
                if out["roofline"]["traffic"] and out["roofline"]["traffic"] > 2.0 * alg_pair * lkp["pairs_per_launch"]:
                    out["roofline"]["traffic_note"] = (
                        "above the algorithmic bytes on purpose: the backward pass leaves its templates (15 KB per feature and "
                        "pair at 21x21) in HBM and the next pair's forward pass fetches them instead of rebuilding them -- "
                        "HBM traffic bought for 9 % fewer vector instructions in a kernel whose HBM fraction is 0.03 "
                        "(DESIGN.md 4.1; ICELK_NO_TEMPLATE_REUSE=1 turns it off)")
                vi = pj.get("lk_fb_valu_insts_per_launch")
                if vi:
                    vi *= scale
                    launch_s = out["roofline"]["avg_launch_us"] * 1e-6
                    rate = vi / launch_s
                    mixes = pj.get("valu_mix") or {}
                    trk_mix = mixes.get("tracker")
                    vio = {"wave_instructions_per_launch": vi,
                           "salu_instructions_per_launch": (pj.get("lk_fb_salu_insts_per_launch") or 0) * scale,
                           "achieved_per_s": rate,
                           # hardware counter of the profiled run: quad-cycles in which a SIMD issued a VALU instruction / all
                           # SIMD cycles of the launch -- a fraction <= 1 by construction
                           "valu_busy_counter_frac": (pj.get("lk_fb_valu_busy_pct") or 0) / 100.0 or None,
                           "lds_bank_conflict_ratio": pj.get("lk_fb_lds_bank_conflict_ratio"),
                           "waves_per_simd": pj.get("lk_fb_waves_per_simd"), "vgprs": pj.get("lk_fb_vgprs"),
                           "sgpr_spills": pj.get("lk_fb_sgpr_spills"), "counters_note": pj.get("counters_note")}
                    if trk_mix:
                        # the peak for THIS instruction mix: measured issue time of each class x its share in the kernel's hot
                        # blocks (profiles/r03_isa_mix_*.json, profiles/valu_class_cost.json)
                        peak = SIMDS / (trk_mix["ns_per_valu_inst"] * 1e-9)
                        vio["mix_weighted"] = {"ns_per_inst_per_simd": trk_mix["ns_per_valu_inst"], "peak_per_s": peak,
                                               "frac": rate / peak, "accounting_error": bool(rate / peak > 1.0),
                                               "hot_class_share": trk_mix["hot_class_share"], "mix_source": trk_mix["source"],
                                               "class_cost_source": mixes.get("class_cost_source")}
                    out["roofline"]["valu_issue"] = vio
                    bes = pj.get("beside_valu_insts_per_launch")
                    if bes and (bes.get("corner_kernel") or bes.get("k_eig_nms")) and trk_mix and cname in ("c2", "c3", "c4") and lkp["pairs_per_launch"] > 1.5:
                        # everything the SIMDs issue per period of the pipeline (two frames: one joint tracker launch, one
                        # corner kernel, two pyramids, one min-distance chain), each kernel priced with its own mix, against
                        # the wall time of that period
                        ns = lambda key: ((mixes.get(key) or trk_mix)["ns_per_valu_inst"])   # noqa: E731
                        parts = {"tracker": (vi, ns("tracker")), "corner_kernel": (bes.get("corner_kernel") or bes.get("k_eig_nms"), ns("corner_kernel")),
                                 "pyramids": (2 * (bes.get("k_pyramid_ahead") or 0), ns("pyramid_one_wave")),
                                 "min_distance_chain": (bes.get("min_distance_chain") or 0, ns("tracker"))}
                        period_s = lkp["pairs_per_launch"] / out["value"]
                        t_min = sum(n * c for n, c in parts.values()) * 1e-9 / SIMDS
                        out["roofline"]["valu_issue"]["pipeline"] = {
                            "wave_instructions_per_period": {k: n for k, (n, _) in parts.items()}, "period_us": period_s * 1e6,
                            "issue_time_at_peak_us": t_min * 1e6, "frac": t_min / period_s,
                            "accounting_error": bool(t_min / period_s > 1.0),
                            "note": "tracker + corner kernel + pyramids + min-distance chain share the SIMDs: time the period's "
                                    "instructions need at the measured issue rate of their classes / wall time of the period"}
            except Exception as exc:
                sys.stderr.write("profiles/pmc_%s.json unreadable: %s\n" % (cname, exc))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.motion)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        code = torch.tensor([exit_code], dtype=torch.int32)
        dist.broadcast(code, src=0)
        exit_code = int(code.item())
        dist.barrier()
        dist.destroy_process_group()
    if ctx is not None:
        ctx.close()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
