#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path (BASELINE.json metric: Msamples/s).

One "step" = one full frame of the workload: RNG seeding + the render megakernel (the reference's timed
region, R/kernel.cu:676-691) and, for N > 1, the single RCCL gather of the row stripes to rank 0.

Default workload = config C2 (BASELINE.json configs[1]): Book-1 final random-spheres scene, list world
("no BVH"), 1200x800, 500 spp, depth 50, fp64 like the reference.  With N GPUs the frame keeps its view
and width but gets N x the rows (1200 x 800N: N x vertical sample density), rows dealt to ranks in
8-row stripes, so every rank renders one C2-frame's worth of pixels of the same distribution: weak scaling.

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene_id, world_kind, width, height, spp, description)
    "c2": (11, 1, 1200, 800, 500, "C2 Book-1 final random-spheres, HittableList world (no BVH), 1200x800x500spp x50 bounces"),
    "c3": (0, 0, 1200, 800, 500, "C3 random-spheres + MovingSphere motion blur, BvhNode world, 1200x800x500spp"),
    "c4": (7, 0, 800, 800, 1000, "C4 Cornell box + 2 rotate/translate instances, 800x800x1000spp"),
    "c5": (9, 0, 1600, 1600, 5000, "C5 Book-2 final scene, 1600x1600x5000spp"),
}

# Algorithmic bytes per element test, fp64: what one test minimally has to read.
BYTES = {"box_tests": 56, "sphere_tests": 36, "msphere_tests": 76, "quad_tests": 132, "xform_entries": 44,
         "medium_calls": 16, "scatters": 32}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # half the 157.3 TF fp32 vector rate


def cpu_baseline(wl, budget_s=12.0):
    """Time the CPU oracle (a port, not the reference itself: the reference cannot be built here) on a
    bounded sample of the same workload: a band of rows through the middle of the frame, few spp."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    scene_id, world, W, H, spp, _ = wl
    orc = Oracle()
    # the GPU box gives one GPU's share of the host: 16 cores (use fewer if the machine has fewer)
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    rows = (H // 2 - 4, H // 2 + 4)
    orc.render(scene_id, world, W, H, 1, rows=rows, threads=cores)  # warm-up: builds the RNG jump table
    t0 = time.time()
    orc.render(scene_id, world, W, H, 1, rows=rows, threads=cores)
    ta = time.time() - t0
    t0 = time.time()
    orc.render(scene_id, world, W, H, 5, rows=rows, threads=cores)  # calibration: per-spp cost without the fixed part
    t1 = max((time.time() - t0 - ta) / 4, 1e-4)
    n_spp = int(max(1, min(spp, budget_s / t1)))
    t0 = time.time()
    _, stats = orc.render(scene_id, world, W, H, n_spp, rows=rows, threads=cores, want_stats=True)
    dt = time.time() - t0
    samples = W * (rows[1] - rows[0]) * n_spp
    bytes_per_ray = sum(BYTES[k] * stats[k] for k in BYTES) / max(stats["rays"], 1)
    return {
        "value": samples / dt * 1e-6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"rows {rows[0]}..{rows[1] - 1} of the {W}x{H} frame at {n_spp} spp ({samples} samples, {dt:.1f} s, "
                  f"OpenMP over 8-pixel chunks, fp64, gcc -O2 -ffp-contract=off)",
        "mray_per_s": stats["rays"] / dt * 1e-6,
    }, bytes_per_ray, stats["rays"] / samples


def measured_traffic(workload, spp, variant):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass (FETCH_SIZE x2 +
    WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes); only valid for the exact configuration profiled."""
    path = os.path.join(ROOT, "profiles", "r01_final_c2_pmc.json")
    if workload != "c2" or spp != 500 or variant != "fast" or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", default="fast", choices=["strict", "fast"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-depth", type=int, default=50, help="diagnostics only (the headline uses the reference's 50)")
    ap.add_argument("--coop-threshold", type=int, default=0, help="tuning knob (0 = library default)")
    ap.add_argument("--flags", type=int, default=0, help="RT_FLAG_* tuning/diagnostic bits")
    ap.add_argument("--shade-batch", type=int, default=0, help="tuning knob (0 = library default)")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="tuning knob: cap resident workgroups per CU")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="after the headline (one frame after another) also time the same K frames with this many in "
                         "flight on per-film streams and report it as \"pipelined\" (1 GPU only; 1 = skip)")
    ap.add_argument("--overdue", type=int, default=0, help="tuning knob: rays/sample budget before a pixel goes cooperative")
    args = ap.parse_args()

    import torch
    import raytracinginoneweekendincuda_amd as rt

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    scene_id, world_kind, W, H0, spp, desc = wl
    if args.spp:
        spp = args.spp
    H = H0 * world                                  # weak scaling: N x the rows, same view
    variant = 0 if args.variant == "strict" else 1

    scene = rt.builtin_scene(scene_id, world_kind, W, H0)     # camera aspect from the base frame
    scene.upload(local_rank)                                  # inputs resident in HBM before timing
    film = rt.Film(W, H, device=local_rank, stripe_rows=8, rank=rank, world_size=world)
    rows_max = max(len(rt.stripe_rows(H, 8, r, world)) for r in range(world))
    mine = torch.zeros(rows_max * W * 3, dtype=torch.float64, device="cuda")
    film.bind_pixels(mine.data_ptr())
    gathered = [torch.empty_like(mine) for _ in range(world)] if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream().cuda_stream
    params = film.params(spp, max_depth=args.max_depth, seed=1984, variant=variant, stream=stream,
                         coop_threshold=args.coop_threshold, overdue=args.overdue, flags=args.flags, shade_batch=args.shade_batch, max_blocks_per_cu=args.blocks_per_cu)

    def step():
        film.launch(scene, params)
        st = film.finish(scene)
        if world > 1:
            dist.gather(mine, gathered, dst=0)      # the one exchange step of the frame
        return st

    for _ in range(args.warmup):
        step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    kernel_s, rays = [], 0
    for _ in range(args.steps):
        st = step()
        kernel_s.append(st.seconds_render)
        rays = st.rays
    fence()
    elapsed = time.perf_counter() - t0
    seed_s = st.seconds_seed

    def pipelined(depth):
        """The same K frames with `depth` of them in flight: film k % depth on its own HIP stream, reaped just before
        its slot is reused.  The next frame's waves occupy the SIMDs the current frame's tail (a few long pixels) leaves
        idle.  Reported beside the headline, never as it."""
        films = [rt.Film(W, H, device=local_rank) for _ in range(depth)]
        plist = [f.params(spp, max_depth=args.max_depth, seed=1984, variant=variant, stream=0,
                          coop_threshold=args.coop_threshold, overdue=args.overdue, flags=args.flags,
                          shade_batch=args.shade_batch, max_blocks_per_cu=args.blocks_per_cu) for f in films]
        busy = [False] * depth

        def run(n):
            for k in range(n):
                if busy[k % depth]:
                    films[k % depth].finish(scene)
                films[k % depth].launch(scene, plist[k % depth])
                busy[k % depth] = True
            for k in range(depth):
                if busy[k]:
                    films[k].finish(scene)
                    busy[k] = False

        run(max(args.warmup, depth))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        return time.perf_counter() - t1

    pipe_elapsed = pipelined(args.pipeline) if (world == 1 and args.pipeline > 1) else None

    t = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_rays = float(tmax[0]), float(tsum[1])
    else:
        total_rays = float(rays)

    if rank == 0:
        samples_per_step = W * H * spp
        value = samples_per_step * args.steps / elapsed * 1e-6
        out = {
            "metric": "Msamples/s (pixels x spp / s), whole job",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc + (f"; {world} GPUs: {W}x{H} (same view, {world}x rows), 8-row stripes round-robin, one RCCL gather"
                                           if world > 1 else ""),
                       "frames_in_flight": 1, "width": W, "height": H, "spp": spp, "max_depth": 50, "seed": 1984, "variant": args.variant,
                       "rng": "XORWOW (cuRAND device-API semantics), one sequence per pixel"},
            "mray_per_s_total": total_rays / (elapsed / args.steps) * 1e-6,
            "mray_per_s_per_gpu": total_rays / world / (elapsed / args.steps) * 1e-6,
            "rays_per_sample": total_rays / samples_per_step,
            "kernel": {"name": "render_kernel", "avg_ms": float(np.mean(kernel_s)) * 1e3, "seed_ms": seed_s * 1e3,
                       "vgprs": st.kernel_vgprs, "lds_bytes": st.lds_bytes},
        }
        if pipe_elapsed is not None:
            out["pipelined"] = {"frames_in_flight": args.pipeline, "value": samples_per_step * args.steps / pipe_elapsed * 1e-6,
                                "unit": "Msamples/s", "ms_per_step": pipe_elapsed / args.steps * 1e3,
                                "note": "same K frames, launched on per-film HIP streams so that frame k+1 fills the SIMDs "
                                        "frame k's tail leaves idle; a throughput mode for frame sequences, not the headline"}
        bytes_per_ray = None
        info_spheres = scene.info()["n_spheres"] if world_kind == 1 else 0
        if world == 1 and not args.no_cpu_baseline:
            base, bytes_per_ray, _ = cpu_baseline(wl)
            out["cpu_baseline"] = base
        if bytes_per_ray is None:
            info = scene.info()
            bytes_per_ray = info["n_spheres"] * BYTES["sphere_tests"] if world_kind == 1 else None
        if bytes_per_ray is not None:
            rays_rank0 = float(rays)
            achieved = bytes_per_ray * rays_rank0 / float(np.mean(kernel_s)) * 1e-9
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.workload, spp, args.variant),
                "algorithmic_bytes_per_ray": bytes_per_ray,
                "note": "algorithmic bytes (SURVEY 8d element sizes x oracle-counted tests per ray) / HIP-event kernel time; "
                        "the tables are chip-resident (scalar cache / L2), so frac may exceed 1: the true limiter is fp64 VALU",
            }
            if world_kind == 1 and info_spheres:
                # the limiter: fp64 VALU.  13 fp64 instructions per ray-sphere test (fast build), 4 cycles per wave64
                # instruction per SIMD, 1024 SIMDs, 2.4 GHz peak clock
                tests = rays_rank0 * info_spheres
                out["roofline"]["valu"] = {
                    "tests_per_s": tests / float(np.mean(kernel_s)),
                    "peak_tests_per_s": 1024 * 2.4e9 * 64 / (13 * 4),
                    "frac": tests / float(np.mean(kernel_s)) / (1024 * 2.4e9 * 64 / (13 * 4)),
                    "unit": "ray-sphere tests/s"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
