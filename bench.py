#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path (BASELINE.json metric: Msamples/s).

One "step" = one full frame of the workload: RNG seeding + the render megakernel (the reference's timed
region, R/kernel.cu:676-691) and, for N > 1, the single RCCL gather of the row stripes to rank 0.

Default workload = config C2 (BASELINE.json configs[1]): Book-1 final random-spheres scene, list world
("no BVH"), 1200x800, 500 spp, depth 50, fp64 like the reference.

Multi-GPU (rows dealt to ranks in 8-row stripes, no data-path collective, one gather at frame end):
  --scaling weak   (default) the frame keeps its view and width and gets N x the rows, so every rank renders one
                   base frame's worth of pixels of the same distribution;
  --scaling strong the BASELINE frame itself is striped over the N ranks (what north_star configs 4-5 do).
  --emulate-ranks N  on ONE GPU: render rank 0..N-1 of an N-way split one after another and report
                   max_r T_r against T_1 / N -- the scaling evidence available without an N-GPU node.

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Besides the contract's keys the JSON line carries `parity` (rows of the frame just timed against the CPU oracle at the
full spp), `roofline` (fp64-VALU issue-slot bound; the contract's HBM algorithmic figure as a secondary key) and
`cpu_baseline` (the oracle timed on the host cores, rank 0 at N = 1 only).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene_id, world_kind, width, height, spp, description)
    "c2": (11, 1, 1200, 800, 500, "C2 Book-1 final random-spheres, HittableList world (no BVH)"),
    "c3": (0, 0, 1200, 800, 500, "C3 random-spheres + MovingSphere motion blur, BvhNode world"),
    "c4": (7, 0, 800, 800, 1000, "C4 Cornell box + 2 rotate/translate instances"),
    "c5": (9, 0, 1600, 1600, 5000, "C5 Book-2 final scene (BVH, Perlin + earth image textures, ConstantMedium volumes)"),
}

# Algorithmic bytes per element test, fp64: what one test minimally has to read.
BYTES = {"box_tests": 56, "sphere_tests": 36, "msphere_tests": 76, "quad_tests": 132, "xform_entries": 44,
         "medium_calls": 16, "scatters": 32}
# Algorithmic fp64 VALU instruction slots per element test (DESIGN.md section 4 has the derivation): the arithmetic the
# reference's test prescribes, with a multiply-add pair counted as ONE slot where the fast build may contract it, a
# divide or a square root as 10 (the gfx950 fp64 sequence: scale, rcp/rsq, Newton steps, fix-up).  Rays carry the
# per-ray fixed work (1/d, d.d, hit record, camera ray share); nothing is counted for integer / RNG / control work.
SLOTS = {"box_tests": 25, "sphere_tests": 13, "msphere_tests": 16, "quad_tests": 19, "xform_entries": 6,
         "medium_draws": 45, "scatters": 40, "noise_calls": 140, "rays": 58}
SLOTS_LIST_WORLD_RAY = 28     # list worlds need no 1/d per ray
SLOTS_FILTERED_SPHERE = 8     # what the sphere-list kernel executes per sphere: the conservative filter (render.hip filter_four)
SLOTS_FILTER_SETUP = 60       # and per ray: 1/|d| (sqrt + divide), foot point, thresholds
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # half the 157.3 TF fp32 vector rate: one wave64 fp64 instruction per 4 cycles per SIMD
PEAK_SLOTS_PER_S = 1024 * 2.4e9 * 64 / 4   # lane-instructions/s: 1024 SIMDs, 2.4 GHz, 16 lanes per cycle

TOL = 1e-5                    # north_star: colour within 1e-5


def earth_bytes():
    """ImageTexture bytes of the reference's earthmap.jpg as its own stb build decodes it (committed fixture)."""
    with np.load(os.path.join(ROOT, "tests", "golden", "earthmap_stb.npz")) as g:
        return np.ascontiguousarray(g["bytes"])


def oracle_band(wl, spp, max_depth, earth, budget_s=14.0):
    """Render a band of full-width rows through the middle of the frame with the CPU oracle at the FULL spp: it is both
    the checker for `parity` and the timed sample for `cpu_baseline` (a port, not the reference itself: the reference
    has no CPU path and cannot be built here).  The band is as many 8-row stripes as fit the time budget."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    scene_id, world, W, H, _, _ = wl
    orc = Oracle()
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    mid = (H // 2) // 8 * 8
    probe_rows = (mid, mid + 1)
    orc.render(scene_id, world, W, H, 1, depth=max_depth, earth=earth, rows=probe_rows, threads=cores)  # builds the RNG jump table
    t0 = time.time()
    orc.render(scene_id, world, W, H, 2, depth=max_depth, earth=earth, rows=probe_rows, threads=cores)
    per_row_spp = max((time.time() - t0) / 2, 1e-5)
    n_rows = int(budget_s / (per_row_spp * spp))
    n_rows = max(1, min(32, n_rows))
    if n_rows >= 8:
        n_rows = n_rows // 8 * 8
    rows = (mid, min(H, mid + n_rows))
    t0 = time.time()
    want, stats = orc.render(scene_id, world, W, H, spp, depth=max_depth, earth=earth, rows=rows, threads=cores, want_stats=True)
    dt = time.time() - t0
    samples = W * (rows[1] - rows[0]) * spp
    base = {
        "value": samples / dt * 1e-6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"rows {rows[0]}..{rows[1] - 1} of the {W}x{H} frame at the full {spp} spp ({samples} samples, {dt:.1f} s, "
                  f"OpenMP over 8-pixel chunks, fp64, gcc -O2 -ffp-contract=off); the same rows are the parity check",
        "mray_per_s": stats["rays"] / dt * 1e-6,
    }
    return base, stats, want[rows[0]:rows[1]], rows


def parity_of(got, want):
    diff = np.abs(got - want)
    q = lambda f: (256.0 * np.clip(f, 0.0, 0.999)).astype(np.int32)   # the PPM writer's quantisation, R/kernel.cu:710-718
    return {
        "within_1e-5": float(np.mean(np.all(diff <= TOL, axis=-1))),
        "bit_exact": float(np.mean(np.all(got.view(np.uint64) == want.view(np.uint64), axis=-1))),
        "ppm8_equal": float(np.mean(np.all(q(got) == q(want), axis=-1))),
        "max_abs_diff": float(diff.max()),
    }


def measured_traffic(workload, spp, variant):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 --pmc pass of this very configuration
    (FETCH_SIZE x2 + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes).  Not measured by this run: the source file
    is named next to the number; None when no profile of this configuration is committed."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("spp", 500 if workload == "c2" else None) == spp and d.get("variant", "fast") == variant and "hbm_bytes_per_launch" in d:
            return d["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def measured_issue(workload, spp, variant):
    """What the SIMDs did during a committed PMC pass of this very configuration: the share of SIMD cycles in which a VALU
    instruction issued (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / kernel time x 2.4 GHz) and the share of live lanes in
    those instructions (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64).  None without such a profile."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        ms = d.get("ms_render_kernels_GRBM_GUI_ACTIVE")
        if d.get("spp") == spp and d.get("variant", "fast") == variant and ms and d.get("SQ_INSTS_VALU"):
            return {"valu_issue_frac": d["SQ_INSTS_VALU"] * 4.0 / (1024 * ms * 1e-3 * 2.4e9),
                    "lane_utilisation": d.get("lane_utilisation"), "valu_instructions_per_frame": d["SQ_INSTS_VALU"],
                    "source": os.path.relpath(path, ROOT),
                    "note": "committed rocprofv3 --pmc pass, not this run: fraction of SIMD cycles with a VALU instruction issuing, "
                            "and live lanes per VALU instruction; their product is the share of the chip's lane-cycles doing "
                            "arithmetic of any kind (fp64, integer RNG, address and control work alike)"}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames (default 3; C5, 31 s a frame: 1)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames first (default 1; C5: 0)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", default="auto", choices=["auto", "strict", "fast"],
                    help="auto = the strict build (the reference's arithmetic, bit-identical frames) for C2 and C3, where it is within "
                         "2-3 per cent of the fast one, and for C5, where 5000 samples per pixel through media give a contracted "
                         "comparison a chance to flip in every pixel; the fast build (FMA contraction) for C4, whose frame equals the "
                         "oracle's bit for bit at the benchmark's own spp all the same")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N GPUs: weak = N x the rows of the base frame (default), strong = the base frame striped over the ranks")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="one GPU: after the headline, render rank 0..N-1 of an N-way stripe split of the frame one after "
                         "another and report max_r T_r vs T_1/N as \"emulated_scaling\"")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (no cpu_baseline, no parity)")
    ap.add_argument("--oracle-budget", type=float, default=14.0, help="seconds of oracle work for the parity / cpu_baseline band")
    ap.add_argument("--max-depth", type=int, default=50, help="diagnostics only (the headline uses the reference's 50)")
    ap.add_argument("--coop-threshold", type=int, default=0, help="tuning knob (0 = library default)")
    ap.add_argument("--flags", type=int, default=0, help="RT_FLAG_* tuning/diagnostic bits")
    ap.add_argument("--shade-batch", type=int, default=0, help="tuning knob (0 = library default)")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="tuning knob: cap resident workgroups per CU")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="after the headline (one frame after another) also time the same K frames with this many in "
                         "flight on per-film streams and report it as \"pipelined\" (1 GPU only; 1 = skip)")
    ap.add_argument("--overdue", type=int, default=0, help="tuning knob: rays/sample budget before a pixel goes cooperative")
    args = ap.parse_args()
    long_frames = args.workload == "c5" and not args.spp   # 31 s a frame at the full 5000 spp
    if args.steps is None:
        args.steps = 1 if long_frames else 3
    if args.warmup is None:
        args.warmup = 0 if long_frames else 1
    if args.pipeline is None:
        args.pipeline = 1 if long_frames else 2

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
    if args.variant == "auto":
        args.variant = "fast" if args.workload == "c4" else "strict"
    scene_id, world_kind, W, H0, spp, desc = wl
    if args.spp:
        spp = args.spp
    H = H0 * world if args.scaling == "weak" else H0   # weak: N x the rows, same view; strong: the frame itself
    variant = 0 if args.variant == "strict" else 1
    earth = earth_bytes() if scene_id in (2, 9) else None

    scene = rt.builtin_scene(scene_id, world_kind, W, H0, earth=earth)   # camera aspect from the base frame
    scene.upload(local_rank)                                  # inputs resident in HBM before timing
    film = rt.Film(W, H, device=local_rank, stripe_rows=8, rank=rank, world_size=world)
    rows_max = max(len(rt.stripe_rows(H, 8, r, world)) for r in range(world))
    mine = torch.zeros(rows_max * W * 3, dtype=torch.float64, device="cuda")
    film.bind_pixels(mine.data_ptr())
    gathered = [torch.empty_like(mine) for _ in range(world)] if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream().cuda_stream
    knobs = dict(coop_threshold=args.coop_threshold, overdue=args.overdue, flags=args.flags, shade_batch=args.shade_batch,
                 max_blocks_per_cu=args.blocks_per_cu)
    params = film.params(spp, max_depth=args.max_depth, seed=1984, variant=variant, stream=stream, **knobs)

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
        kernel_s.append(st.seconds_render)   # HIP events recorded on the stream the kernel is launched on
        rays = st.rays
    fence()
    elapsed = time.perf_counter() - t0
    seed_s = st.seconds_seed

    def pipelined(depth):
        """The same K frames with `depth` of them in flight: film k % depth on its own HIP stream, reaped just before
        its slot is reused.  The next frame's waves occupy the SIMDs the current frame's tail (a few long pixels) leaves
        idle.  Reported beside the headline, never as it."""
        films = [rt.Film(W, H, device=local_rank) for _ in range(depth)]
        plist = [f.params(spp, max_depth=args.max_depth, seed=1984, variant=variant, stream=0, **knobs) for f in films]
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

    def emulate(n_ranks):
        """Rank r of an n_ranks-way stripe split of the BASELINE frame, one rank after another on this one GPU."""
        per_rank = []
        for r in range(n_ranks):
            f = rt.Film(W, H0, device=local_rank, stripe_rows=8, rank=r, world_size=n_ranks)
            p = f.params(spp, max_depth=args.max_depth, seed=1984, variant=variant, **knobs)
            f.render(scene, p.samples_per_pixel, max_depth=args.max_depth, variant=variant, **knobs)   # warm-up
            ts = []
            for _ in range(3):
                s_ = f.render(scene, p.samples_per_pixel, max_depth=args.max_depth, variant=variant, **knobs)
                ts.append(s_.seconds_seed + s_.seconds_render)
            per_rank.append(float(np.min(ts)) * 1e3)   # best of three: a rank's frame is short, the first one after a switch of films often slow
            del f
        return per_rank

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
        kernel_avg = float(np.mean(kernel_s))
        workload = f"{desc}, {W}x{H0}x{spp}spp x{args.max_depth} bounces"
        if world > 1:
            workload += (f"; {world} GPUs, weak scaling: {W}x{H} (same view, {world}x rows)" if args.scaling == "weak"
                         else f"; {world} GPUs, strong scaling: the {W}x{H} frame itself") + ", 8-row stripes round-robin, one RCCL gather"
        out = {
            "metric": "Msamples/s (pixels x spp / s), whole job",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "frames_in_flight": 1, "width": W, "height": H, "spp": spp,
                       "max_depth": args.max_depth, "seed": 1984, "variant": args.variant,
                       "rng": "XORWOW (cuRAND device-API semantics), one sequence per pixel"},
            "mray_per_s_total": total_rays / (elapsed / args.steps) * 1e-6,
            "mray_per_s_per_gpu": total_rays / world / (elapsed / args.steps) * 1e-6,
            "rays_per_sample": total_rays / samples_per_step,
            "kernel": {"name": "render_kernel", "avg_ms": kernel_avg * 1e3, "seed_ms": seed_s * 1e3,
                       "vgprs": st.kernel_vgprs, "lds_bytes": st.lds_bytes, "kind": st.kernel_kind},
        }
        if pipe_elapsed is not None:
            out["pipelined"] = {"frames_in_flight": args.pipeline, "value": samples_per_step * args.steps / pipe_elapsed * 1e-6,
                                "unit": "Msamples/s", "ms_per_step": pipe_elapsed / args.steps * 1e3,
                                "note": "same K frames, launched on per-film HIP streams so that frame k+1 fills the SIMDs "
                                        "frame k's tail leaves idle; a throughput mode for frame sequences, not the headline"}
        if world == 1 and args.emulate_ranks > 1:
            per_rank = emulate(args.emulate_ranks)
            t1_ms = kernel_avg * 1e3 + seed_s * 1e3
            out["emulated_scaling"] = {
                "ranks": args.emulate_ranks, "scaling": "strong", "frame": f"{W}x{H0}x{spp}spp", "t1_ms": t1_ms,
                "per_rank_ms": per_rank, "max_rank_ms": max(per_rank),
                "speedup": t1_ms / max(per_rank), "efficiency": t1_ms / max(per_rank) / args.emulate_ranks,
                "note": "rank r's stripes rendered alone on this one GPU (seed + render kernels, HIP events); an N-GPU run "
                        "finishes with its slowest rank plus one gather of a few MB"}
        if world == 1 and not args.no_cpu_baseline:
            base, stats, want, rows = oracle_band(wl, spp, args.max_depth, earth, args.oracle_budget)
            out["cpu_baseline"] = base
            # ---- parity: the rows of the frame just timed against the oracle at the full spp ----
            frame = film.download()
            par = {"variant": args.variant, "rows": [rows[0], rows[1] - 1], "width": W, "spp": spp, "tolerance": TOL,
                   "checker": "oracle/rtow_oracle.c (CPU restatement; parity unpinned, see DESIGN.md section 2)"}
            par.update(parity_of(frame[rows[0]:rows[1]], want))
            # the other build on the same rows: each 8-row stripe rendered on its own through the stripe partition
            other = 1 - variant
            n_stripes = (H + 7) // 8
            got_other = np.zeros_like(want)
            for k in range(rows[0] // 8, (rows[1] + 7) // 8):
                f2 = rt.Film(W, H, device=local_rank, stripe_rows=8, rank=k, world_size=n_stripes)
                f2.render(scene, spp, max_depth=args.max_depth, variant=other)
                band = f2.download()[k * 8:min(H, k * 8 + 8)]
                lo, hi = max(rows[0], k * 8), min(rows[1], k * 8 + 8)
                got_other[lo - rows[0]:hi - rows[0]] = band[lo - k * 8:hi - k * 8]
            par["other_variant"] = dict(variant="strict" if other == 0 else "fast", **parity_of(got_other, want))
            out["parity"] = par
            # ---- roofline: fp64 VALU issue slots (the limiter); the contract's HBM algorithmic figure beside it ----
            n_rays = max(stats["rays"], 1)
            slots = dict(SLOTS)
            if world_kind == 1:
                slots["rays"] = SLOTS_LIST_WORLD_RAY
            slots_per_ray = sum(slots[k] * stats[k] for k in slots) / n_rays
            bytes_per_ray = sum(BYTES[k] * stats[k] for k in BYTES) / n_rays
            slots_per_s = slots_per_ray * float(rays) / kernel_avg
            achieved_tf = slots_per_s * 2 * 1e-12     # issue-slot equivalent: the peak counts every slot as an FMA (2 flop)
            traffic, traffic_src = measured_traffic(args.workload, spp, args.variant)
            out["roofline"] = {
                "bound": "valu_fp64", "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                "traffic_source": traffic_src if traffic is not None else "not measured in this run; no committed PMC profile of this configuration",
                "slots_per_ray": slots_per_ray, "slots_per_s": slots_per_s, "peak_slots_per_s": PEAK_SLOTS_PER_S,
                "tests_per_ray": {k: stats[k] / n_rays for k in slots if k != "rays"},
                "executed": None,
                "issue": measured_issue(args.workload, spp, args.variant),
                "note": "achieved = algorithmic fp64 VALU instruction slots per ray (oracle-counted element tests x the slot table "
                        "in bench.py / DESIGN.md) x rays of the timed launch / HIP-event kernel time, expressed at 2 flop per slot "
                        "against the 78.6 TFLOP/s vector-fp64 peak (one wave64 instruction per 4 cycles per SIMD at 2.4 GHz)",
                "hbm_algorithmic": {
                    "achieved": bytes_per_ray * float(rays) / kernel_avg * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": bytes_per_ray * float(rays) / kernel_avg * 1e-9 / HBM_PEAK_GBS, "bytes_per_ray": bytes_per_ray,
                    "note": "SURVEY 8d definition (element sizes x oracle-counted tests per ray / kernel time); the tables are "
                            "chip-resident (scalar cache / LDS / L2), so this is not a bound and may exceed 1"},
            }
            if args.workload == "c2" and not (args.flags & 256):
                ex = dict(slots)
                ex["sphere_tests"] = SLOTS_FILTERED_SPHERE
                ex["rays"] = slots["rays"] + SLOTS_FILTER_SETUP
                ex_per_ray = sum(ex[k] * stats[k] for k in ex) / n_rays
                out["roofline"]["executed"] = {
                    "slots_per_ray": ex_per_ray, "frac": ex_per_ray * float(rays) / kernel_avg / PEAK_SLOTS_PER_S,
                    "note": "the same ratio with the slots the kernel issues instead of the reference's arithmetic: every sphere of "
                            "the list goes through an 8-instruction conservative filter (13 for the reference's discriminant and "
                            "compare), the few survivors per ray through the reference's test; frac above counts the reference's 13"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
