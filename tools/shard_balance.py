"""Kernel time of every rank's share of the C2 frame at world sizes 1, 2, 4, 8, measured one share after the other on ONE GPU
(no exchange): max over ranks / (full-frame time / N) is the load-balance + tail efficiency the tile round-robin can reach before
any RCCL cost.  Run on the GPU box from the repo root."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as entry
hjr = entry.load_package()
from scene_util import Cornell

s = Cornell("render_option_c2.json")
d = s.device()
W, H, SPP = 1920, 1080, 256
fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
full = None
for world in (1, 2, 4, 8):
    times = []
    for rank in range(world):
        p = s.hjr_params(W, H, SPP, rank=rank, world_size=world, flags=hjr.FLAG_ZERO_UNOWNED if world > 1 else 0)
        for rep in range(2):
            d.render_device(p, fb.data_ptr(), None, None, stream)
            torch.cuda.synchronize()
        times.append(d.stats()["last_kernel_ms"])
    if world == 1:
        full = times[0]
    print("world %d: kernel ms per rank min %.2f max %.2f mean %.2f | ideal %.2f | efficiency of the slowest rank %.3f" %
          (world, min(times), max(times), sum(times) / len(times), full / world, full / world / max(times)))
d.close()
