#!/usr/bin/env python3
"""k_bgr2gray alone on the device, 30 conversions of a 4000x3000 BGR frame resident in HBM (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iceberg_tracking_code_amd import Context
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4000, 3000)
ctx = Context(w, h, n_slots=4, max_pts=1024)
rgb = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for rep in range(30):
    ctx.cvt_bgr_device(rep % 4, rgb.data_ptr(), w, h, 3 * w)
ctx.sync()
ctx.close()
