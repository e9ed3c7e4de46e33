"""The N > 1 path on CPU: world_size-2 (and 3) gloo processes each own their row stripes, fill their
compact buffer, one gather to rank 0, de-interleave -> the full frame.  The per-rank "renderer" here is the
CPU oracle restricted to the rank's rows (tests may use it); on the GPU box the same plumbing runs over
RCCL with the HIP kernel (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, STRIPE = 24, 37, 1, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracinginoneweekendincuda_amd as rt
    from conftest import Oracle
    orc = Oracle()
    rows = rt.stripe_rows(H, STRIPE, rank, world)
    rows_max = max(len(rt.stripe_rows(H, STRIPE, r, world)) for r in range(world))
    mine = torch.zeros(rows_max * W * 3, dtype=torch.float64)
    # render only the rows this rank owns (contiguous runs of the stripe pattern)
    compact = []
    for j in rows:
        fb = orc.render(10, 0, W, H, SPP, rows=(j, j + 1), threads=1)
        compact.append(fb[j])
    if compact:
        flat = np.concatenate(compact).ravel()
        mine[: flat.size] = torch.from_numpy(flat)
    gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    if rank == 0:
        g = torch.stack(gathered).numpy()
        frame = rt.deinterleave(g, W, H, STRIPE, world)
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stripes_gather_deinterleave(tmp_path, oracle, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    want = oracle.render(10, 0, W, H, SPP)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
