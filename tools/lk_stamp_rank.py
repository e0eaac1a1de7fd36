import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 3)
n = int((a[:,0] != 0).sum())
d = (a[:,1].astype(np.int64) - a[:,0].astype(np.int64))
idx = np.nonzero(a[:,0] != 0)[0]
d = d[idx]
print("workgroups", n, "sum/4096 slots (us @2.4GHz)", d.sum()/4096/2400.0, "max (us)", d.max()/2400.0, "median", np.median(d)/2400.0)
dec = np.array_split(np.arange(len(d)), 10)
print("mean duration (us) by decile of workgroup index:", [round(float(d[i].mean())/2400.0,1) for i in dec])
print("share of the 5% slowest by decile:", [int((d[i] >= np.percentile(d,95)).sum()) for i in dec])
