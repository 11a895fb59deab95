"""The N>1 path on CPU: ranks over gloo run the same tile-shard + framebuffer-exchange logic bench.py runs over RCCL.
The per-rank shard comes from the oracle (the HIP kernel needs a GPU); what is under test is the partition and the
collectives: both the full-frame SUM reduce and the owned-tiles gather must reproduce the single-process frame bit for bit."""
import os
import subprocess
import sys

import numpy as np

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["HJR_ROOT"], "tests"))
import oracle_binding as ob
from scene_util import Cornell, hjr
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
c = Cornell()
w, h, spp = 75, 41, 2   # ragged: 10 x 6 tiles, the last column / row partly outside the frame
full, _, _, _ = ob.OracleScene(c.arrays, ob.MATH_PORTABLE).render(c.oracle_params(w, h, spp), nthreads=2, want_aovs=False)
mask = hjr.owned_tile_mask(w, h, rank, world)
shard = np.where(mask[..., None], full, np.float32(0)).astype(np.float32)   # what HJR_FLAG_ZERO_UNOWNED produces
fb = torch.from_numpy(shard.copy())
hjr.exchange_framebuffer(fb, dst=0)
ok = True
if rank == 0:
    ok = np.array_equal(fb.numpy().view(np.uint32), full.view(np.uint32))
# the exchange sized by ownership (what bench.py and henjou_cli run over RCCL): each rank contributes only its packed tiles
n_max = hjr.owned_tiles(w, h, 0, world)
mine = hjr.pack_tiles(full, rank, world)          # what HJR_FLAG_PACKED makes the kernel write
assert mine.shape[0] == hjr.owned_tiles(w, h, rank, world) and int(np.sum(mask)) <= mine.shape[0] * 64
padded = np.zeros((n_max, 64, 4), np.float32)
padded[:mine.shape[0]] = mine
frame = hjr.gather_tiles(torch.from_numpy(padded), w, h, dst=0)
if rank == 0:
    ok = ok and np.array_equal(frame.numpy().view(np.uint32), full.view(np.uint32))
    open(os.environ["HJR_OUT"], "w").write("OK" if ok else "MISMATCH")
dist.barrier()
dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("nproc", [2, 3])
def test_tile_shard_reduce_and_gather(tmp_path, nproc):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "result.txt"
    env = dict(os.environ, HJR_ROOT=root, HJR_OUT=str(out), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1",
           "--master-port", str(29571 + nproc), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert out.read_text() == "OK"
