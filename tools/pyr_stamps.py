import numpy as np, sys
a=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,8).astype(np.int64)
d=np.diff(a[:,:6],axis=1)
print("workgroups",len(a),"median cycles per stage [load, level1, copy1, level2, copy2+level3+copy3]:",np.median(d,axis=0).tolist())
print("p90:",np.percentile(d,90,axis=0).tolist(), "total median",np.median(a[:,5]-a[:,0]), "max", (a[:,5]-a[:,0]).max())
