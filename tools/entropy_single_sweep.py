import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rustyhgi_amd as H
from rustyhgi_amd import entropy
from oracle import hgi_oracle as O
img = O.synth(O.SYNTH_RAMP, 0x48474933, 0, 4096, 4096)
grid = O.encode(img, 4, O.linear_lut(2)[0])
d = torch.from_numpy(grid).cuda()
for _ in range(5): s = entropy.deflate_grid(d)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); s = entropy.deflate_grid(d); ts.append(time.perf_counter() - t0)
print("wgs/cu=%s: median %.3f ms min %.3f" % (os.environ.get("HGI_ENTROPY_WGS_PER_CU", "auto"), sorted(ts)[10] * 1e3, min(ts) * 1e3))
