"""The N > 1 path on CPU.  bench.py's own multi-rank code -- bench.frames_leg: fences, K frames, the one gather per frame
(raytracinginoneweekendincuda_amd.stripes.StripeExchange), the reductions over ranks, rank 0's de-interleave -- runs here with
world_size 2 and 3 over gloo; only the renderer is a stand-in: the CPU oracle restricted to the rank's rows (tests may use it)
instead of the HIP kernel.  On the GPUs the same function runs over RCCL (bench.py main()).  Also: `bench.py --gpus N`
without a launcher starts its own ranks (--rendezvous-only: the plumbing without a renderer)."""
import json
import os
import socket
import subprocess
import sys
import time

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
    import bench
    from conftest import Oracle
    orc = Oracle()
    calls = []

    def make_renderer(ex):
        assert ex.rows == [j for j in range(H) if (j // STRIPE) % world == rank]

        def render():   # the stand-in for film.launch + film.finish: this rank's rows, compact, into the exchange's buffer
            t0 = time.perf_counter()
            ex.mine.zero_()
            compact = [orc.render(10, 0, W, H, SPP, rows=(j, j + 1), threads=1)[j] for j in ex.rows]
            if compact:
                flat = np.concatenate(compact).ravel()
                ex.mine[: flat.size] = torch.from_numpy(flat)
            calls.append(1)
            return time.perf_counter() - t0, len(ex.rows) * W * SPP, None
        return render

    out = bench.frames_leg(dist, rank, world, "cpu", W, H, steps=2, warmup=1, make_renderer=make_renderer, stripe_rows=STRIPE)
    assert len(calls) == 3                                     # one warm-up frame, two timed ones
    assert len(out["per_rank_render_ms"]) == world and out["elapsed"] > 0 and out["gather_ms"] >= 0
    assert out["rays_total"] == W * H * SPP                    # summed over ranks: every row rendered exactly once
    if rank == 0:
        np.save(out_path, out["frame"])
    else:
        assert out["frame"] is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_frames_leg_over_gloo(tmp_path, oracle, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    want = oracle.render(10, 0, W, H, SPP)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (the form the driver uses): bench.py starts the two ranks itself, they find
    each other over 127.0.0.1, rank 0's stdout is the caller's.  --rendezvous-only stops before anything needs a GPU."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_ranks"] == 2 and line["sum_of_rank_plus_one"] == 3.0 and line["local_rank_env"] == "0"


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="only meaningful on a machine with fewer than two GPUs")
def test_bench_without_enough_gpus_says_so_before_starting_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs 2 GPUs" in (r.stderr + r.stdout)


def test_a_launcher_that_disagrees_with_gpus_is_refused():
    env = dict(_clean_env(), WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)
