import sys
sys.path.insert(0, "/root/repo")
from iceberg_tracking_code_amd import Context
w, h = 4000, 3000
ctx = Context(w, h, n_slots=2, max_pts=1 << 16)
ctx.synth_frame(0, w, h, 10, -20, 1234)
for q in (0.007,):
    c = ctx.good_features(0, 10000, q, 10, False, 10)
    print(q, len(c), ctx.detect_stats(), ctx.detect_fast_stats(w, h))
ctx.close()
