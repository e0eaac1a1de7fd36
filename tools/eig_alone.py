import sys
sys.path.insert(0, "/root/repo")
from iceberg_tracking_code_amd import Context
w, h = 4000, 3000
ctx = Context(w, h, n_slots=2, max_pts=1 << 16)
ctx.synth_frame(0, w, h, 10, -20, 1234)
ctx.seg_detect(0, 10000, 0.007, 10, False, 10)    # prepared candidates are cut at the quality level of the latest detection
for bs in (10, 3, 5, 7):
    ctx.seg_detect_prepare(0, False, bs); ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(6):
        ctx.synth_frame(0, w, h, 10, -20, 1234)    # new generation: the prepared candidates are stale
        ctx.seg_detect_prepare(0, False, bs)
    ctx.sync(); ctx.prof_enable(False)
    print("blockSize", bs, "corner kernel alone: %.1f us" % ctx.prof_table()["corner_candidates"]["avg_us"])
ctx.close()
