"""Per-tile map of the pyramid kernel's workgroup lifetimes (hundreds of cycles) from an ICELK_PYR_STAMPS file."""
import sys
import numpy as np
f, gx, gy = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
a = np.fromfile(f, dtype=np.uint64).reshape(-1, 8).astype(np.int64)[:gx * gy]
A = a.reshape(gy, gx, 8)
life = A[:, :, 5] - A[:, :, 0]
edge = np.zeros((gy, gx), bool); edge[0] = edge[-1] = True; edge[:, 0] = edge[:, -1] = True
print("life: interior median %d max %d, edge median %d max %d" % (np.median(life[~edge]), life[~edge].max(), np.median(life[edge]), life[edge].max()))
for k, name in enumerate(["load+fill0", "level1", "copy1", "fill1+level2", "copy2+fill2+level3+copy3"], 1):
    d = A[:, :, k] - A[:, :, k - 1]
    print("%-26s edge median %5d   interior median %5d p90 %5d max %5d" % (name, np.median(d[edge]), np.median(d[~edge]), np.percentile(d[~edge], 90), d[~edge].max()))
np.set_printoptions(linewidth=250)
if "-m" in sys.argv:
    print((life / 100).astype(int))
