#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path (BASELINE.json metric: Msamples/s).

One "step" = one full frame of the workload: RNG seeding + the render megakernel (the reference's timed
region, R/kernel.cu:676-691) and, for N > 1, the single RCCL gather of the row stripes to rank 0.

Default workload = config C2 (BASELINE.json configs[1]): Book-1 final random-spheres scene, list world
("no BVH"), 1200x800, 500 spp, depth 50, fp64 like the reference.

Multi-GPU (rows dealt to ranks in 8-row stripes, no data-path collective, one gather at frame end).  Both legs are
measured in one run and reported in one line:
  weak    (the headline, "scaling": "weak") the frame keeps its view and width and gets N x the rows, so every rank
          renders one base frame's worth of pixels of the same distribution;
  strong  (the "strong" sub-record) the BASELINE frame itself is striped over the N ranks (what north_star configs 4-5 do),
          with every rank's render time and the gather time.
  --scaling strong swaps which of the two is the headline.
  --emulate-ranks N  on ONE GPU: render rank 0..N-1 of an N-way split one after another and report
          max_r T_r against T_1 / N -- the scaling evidence available without an N-GPU node.

    python bench.py                      # 1 GPU
    python bench.py --gpus N             # starts its own N ranks (one process per GPU, RCCL) when no launcher did
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Besides the contract's keys the JSON line carries `parity` (rows of the frame just timed against the CPU oracle at the
full spp), `roofline` (fp64-VALU issue-slot bound, scalars only) and `cpu_baseline` (the oracle timed on the host cores,
rank 0 at N = 1 only).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
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
SLOTS_FILTERED_SPHERE = 4.5   # what the sphere-list kernel executes per sphere: the conservative filter in packed fp32, 7 v_pk_* + 2 compares per
                              # PAIR of spheres (render.hip filter_pairs; each issues in the slot of one fp64 instruction)
SLOTS_FILTERED_SPHERE_FP64 = 8  # ... and its fp64 form, one sphere at a time (RT_FLAG_FILTER_FP64; render.hip filter_four)
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


def committed_pmc(workload, spp, variant):
    """The newest committed rocprofv3 --pmc record of this very configuration (profiles/rNN_<workload>_pmc.json), or None."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("spp", 500 if workload == "c2" else None) == spp and d.get("variant", "fast") == variant:
            return d, os.path.relpath(path, ROOT)
    return None, None


def committed_executed(workload, kernel_kind, variant):
    """Executed fp64-slot count per ray of a kernel that does not make the reference's element tests one for one, as measured with
    the RT_PHASES build and committed (profiles/rNN_<workload>_executed.json); None when there is none for this kernel."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_executed.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("kernel_kind") == kernel_kind and d.get("variant") == variant and "slots_per_ray_executed" in d:
            return d, os.path.relpath(path, ROOT)
    return None


def roofline_of(workload, world_kind, spp, variant, flags, stats, rays, kernel_s, kernel_kind=-1):
    """The `roofline` object: scalars only, so that a record which keeps one level of the line keeps all of it.
    `frac` counts the fp64 VALU slots the kernel EXECUTES (sphere-list kernel: the 8-instruction filter, not the reference's
    13-instruction discriminant); `frac_reference_arithmetic` counts the reference's arithmetic for the same element tests."""
    n_rays = max(stats["rays"], 1)
    slots = dict(SLOTS)
    if world_kind == 1:
        slots["rays"] = SLOTS_LIST_WORLD_RAY
    ref_per_ray = sum(slots[k] * stats[k] for k in slots) / n_rays
    ex = dict(slots)
    executed_note = "the kernel executes the reference's arithmetic for every element test it makes"
    if workload == "c2" and not (flags & 256):
        ex["sphere_tests"] = SLOTS_FILTERED_SPHERE_FP64 if (flags & 2048) else SLOTS_FILTERED_SPHERE
        ex["rays"] = slots["rays"] + SLOTS_FILTER_SETUP
        executed_note = ("every sphere of the list goes through the conservative filter -- 9 instructions per two spheres in packed fp32 "
                         "(8 per sphere in fp64 with RT_FLAG_FILTER_FP64; 13 for the reference's discriminant and compare) -- the few "
                         "survivors per ray through the reference's test")
    ex_per_ray = sum(ex[k] * stats[k] for k in ex) / n_rays
    executed_src = "slot table x oracle-counted element tests"
    measured = committed_executed(workload, kernel_kind, variant)
    if measured is not None:   # kernels that do not make the reference's element tests one for one (library tree, segmented walk)
        ex_per_ray = measured[0]["slots_per_ray_executed"]
        executed_src = measured[1]
        executed_note = ("the kernel walks the library's tree, not the reference's: its node visits and leaf tests per ray were counted "
                         "with the RT_PHASES build (" + measured[1] + "), fp32 slab tests at half an fp64 slot per operation")
    bytes_per_ray = sum(BYTES[k] * stats[k] for k in BYTES) / n_rays
    rays_per_s = float(rays) / kernel_s
    pmc, pmc_src = committed_pmc(workload, spp, variant)
    traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
    issue = lanes = None
    if pmc and pmc.get("ms_render_kernels_GRBM_GUI_ACTIVE") and pmc.get("SQ_INSTS_VALU"):
        issue = pmc["SQ_INSTS_VALU"] * 4.0 / (1024 * pmc["ms_render_kernels_GRBM_GUI_ACTIVE"] * 1e-3 * 2.4e9)
        lanes = pmc.get("lane_utilisation")
    hbm_alg = bytes_per_ray * rays_per_s * 1e-9
    return {
        "bound": "valu_fp64", "unit": "TFLOP/s", "peak": FP64_VECTOR_PEAK_TFLOPS,
        "achieved": ex_per_ray * rays_per_s * 2 * 1e-12,       # issue-slot equivalent: the peak counts every slot as an FMA (2 flop)
        "frac": ex_per_ray * rays_per_s / PEAK_SLOTS_PER_S,
        "slots_per_ray": ex_per_ray, "executed_source": executed_src,
        "frac_reference_arithmetic": ref_per_ray * rays_per_s / PEAK_SLOTS_PER_S,
        "achieved_reference_arithmetic": ref_per_ray * rays_per_s * 2 * 1e-12,
        "slots_per_ray_reference_arithmetic": ref_per_ray,
        "valu_issue_frac": issue, "lane_utilisation": lanes,
        "issue_source": pmc_src if issue is not None else "no committed PMC profile of this configuration",
        "hbm_algorithmic_frac": hbm_alg / HBM_PEAK_GBS, "hbm_algorithmic_gbs": hbm_alg, "hbm_peak_gbs": HBM_PEAK_GBS,
        "bytes_per_ray": bytes_per_ray,
        "traffic": traffic, "traffic_bytes": traffic,
        "traffic_source": pmc_src if traffic is not None else "not measured in this run; no committed PMC profile of this configuration",
        "note": "frac = fp64 VALU slots executed per ray (oracle-counted element tests x the slot table in bench.py / DESIGN.md; "
                + executed_note + ") x rays of the timed launch / HIP-event kernel time / 3.93e13 lane-slots/s (one wave64 fp64 "
                "instruction per 4 cycles per SIMD, 1024 SIMDs, 2.4 GHz; x2 flop = 78.6 TFLOP/s); frac_reference_arithmetic = the same "
                "with the reference's arithmetic for every test; valu_issue_frac / lane_utilisation / traffic come from the committed "
                "rocprofv3 --pmc record named beside them, not from this run; hbm_algorithmic_* is the SURVEY 8d figure (tables are "
                "chip-resident: not a bound, may exceed 1)",
    }


# ------------------------------------------------------------------------------------------------------------------------
# frames on N ranks.  Used by main() with the HIP renderer over RCCL and by tests/test_dist_gloo.py with a CPU stand-in
# renderer over gloo: the fences, the exchange and the reductions are this function either way.
# ------------------------------------------------------------------------------------------------------------------------
def frames_leg(dist, rank, world, device, width, height, steps, warmup, make_renderer, stripe_rows=8):
    """`steps` timed frames of a width x height frame striped over `world` ranks: every rank renders its stripes into its
    compact buffer, one gather per frame brings them to rank 0.  make_renderer(exchange) returns render() -> (seconds of
    this rank's kernels, rays traced, stats).  Timed exactly as the task contract says: warm-up frames, barrier + device
    sync, K frames, barrier + device sync, MAX over ranks.  Returns a dict (on every rank; `frame` on rank 0 only)."""
    import torch
    from raytracinginoneweekendincuda_amd.stripes import StripeExchange
    on_gpu = torch.device(device).type == "cuda"
    ex = StripeExchange(dist, width, height, stripe_rows, rank, world, device)
    render = make_renderer(ex)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def fence():
        if world > 1:
            dist.barrier()
        sync()

    for _ in range(warmup):
        render()
        ex.gather()
    fence()
    t0 = time.perf_counter()
    render_s, gather_s, rays, stats, stats_all = [], [], 0, None, []
    for _ in range(steps):
        secs, rays, stats = render()          # returns when this rank's kernels are done (HIP events on the launch stream)
        render_s.append(secs)
        stats_all.append(stats)
        g0 = time.perf_counter()
        ex.gather()                           # the one exchange step of the frame
        sync()
        gather_s.append(time.perf_counter() - g0)
    fence()
    elapsed = time.perf_counter() - t0
    mine = torch.tensor([elapsed, float(np.mean(render_s)) if render_s else 0.0, float(np.mean(gather_s)) if gather_s else 0.0,
                         float(rays)], dtype=torch.float64, device=device)
    if world > 1:
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).cpu().numpy()
    else:
        every = mine.cpu().numpy()[None]
    return {
        "elapsed": float(every[:, 0].max()), "per_rank_render_ms": [float(x) * 1e3 for x in every[:, 1]],
        "gather_ms": float(every[:, 2].max()) * 1e3, "rays_total": float(every[:, 3].sum()),
        "render_s": render_s, "stats": stats, "stats_all": stats_all, "exchange": ex, "frame": ex.frame() if rank == 0 else None,
    }


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def start_own_ranks(n, need_gpus=True):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, BEFORE this process
    touches a GPU (a process that has initialised HIP must not be replaced or forked into ranks).  Rank 0's stdout -- the
    JSON line -- is ours; the exit code is the worst of the ranks'."""
    import torch  # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if need_gpus and have < n:
        raise SystemExit(f"bench.py --gpus {n} needs {n} GPUs on this node: found {have} (the render path has no CPU fallback)")
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    try:
        for p in procs:
            worst = max(worst, abs(p.wait()))
    finally:
        for p in procs:   # a rank that died leaves the others waiting at a barrier: end exactly the processes started here
            if p.poll() is None:
                p.terminate()
    raise SystemExit(worst)


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
                    help="N GPUs: which leg is the headline (the other one is reported as a sub-record): weak = N x the rows of the "
                         "base frame (default), strong = the base frame striped over the ranks")
    ap.add_argument("--one-leg", action="store_true", help="N GPUs: measure the headline leg only")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="one GPU: after the headline, render rank 0..N-1 of an N-way stripe split of the frame one after "
                         "another and report max_r T_r vs T_1/N as \"emulated_scaling\"")
    ap.add_argument("--emulate-ppw", default="", help="with --emulate-ranks: comma-separated pixels_per_wave values to repeat the "
                                                     "emulation with (tuning; the record of record uses the library's own choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (no cpu_baseline, no parity)")
    ap.add_argument("--oracle-budget", type=float, default=14.0, help="seconds of oracle work for the parity / cpu_baseline band")
    ap.add_argument("--max-depth", type=int, default=50, help="diagnostics only (the headline uses the reference's 50)")
    ap.add_argument("--coop-threshold", type=int, default=0, help="tuning knob (0 = library default)")
    ap.add_argument("--flags", type=int, default=0, help="RT_FLAG_* tuning/diagnostic bits")
    ap.add_argument("--shade-batch", type=int, default=0, help="tuning knob (0 = library default)")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="tuning knob: cap resident workgroups per CU")
    ap.add_argument("--pixels-per-wave", type=int, default=0, help="rt_render_params.pixels_per_wave (0 = the library's choice)")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="after the headline (one frame after another) also time the same K frames with this many in "
                         "flight on per-film streams and report it as \"pipelined\" (1 GPU only; 1 = skip)")
    ap.add_argument("--overdue", type=int, default=0, help="tuning knob: rays/sample budget before a pixel goes cooperative")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, form the process group over gloo on the CPU, reduce one number, print it and exit: checks "
                         "the launcher plumbing (also on a machine without GPUs); renders nothing")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        start_own_ranks(args.gpus, need_gpus=not args.rendezvous_only)      # never returns
    if args.rendezvous_only:
        import torch
        import torch.distributed as dist
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_ranks": world, "sum_of_rank_plus_one": float(t[0]),
                              "local_rank_env": os.environ.get("LOCAL_RANK", "0")}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or without any launcher (bench.py then starts its own ranks)")
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
    variant = 0 if args.variant == "strict" else 1
    earth = earth_bytes() if scene_id in (2, 9) else None

    scene = rt.builtin_scene(scene_id, world_kind, W, H0, earth=earth)   # camera aspect from the base frame
    scene.upload(local_rank)                                  # inputs resident in HBM before timing
    stream = torch.cuda.current_stream().cuda_stream
    knobs = dict(coop_threshold=args.coop_threshold, overdue=args.overdue, flags=args.flags, shade_batch=args.shade_batch,
                 max_blocks_per_cu=args.blocks_per_cu, pixels_per_wave=args.pixels_per_wave)
    films = {}

    def hip_renderer(height):
        """This rank's stripes of a W x height frame through the C-ABI, rendered straight into the exchange's buffer."""
        def make(ex):
            film = rt.Film(W, height, device=local_rank, stripe_rows=ex.stripe, rank=rank, world_size=world)
            film.bind_pixels(ex.mine.data_ptr())
            params = film.params(spp, max_depth=args.max_depth, seed=1984, variant=variant, stream=stream, **knobs)
            films[height] = film

            def render():
                film.launch(scene, params)
                st = film.finish(scene)
                return st.seconds_seed + st.seconds_render, st.rays, st
            return render
        return make

    def leg(scaling):
        height = H0 * world if scaling == "weak" else H0   # weak: N x the rows, same view; strong: the frame itself
        out = frames_leg(dist, rank, world, torch.device("cuda", local_rank), W, height, args.steps, args.warmup, hip_renderer(height))
        out["height"] = height
        return out

    head = leg(args.scaling)
    other = None
    if world > 1 and not args.one_leg:
        other = leg("strong" if args.scaling == "weak" else "weak")
    H = head["height"]
    film = films[H]
    st = head["stats"]
    elapsed, total_rays = head["elapsed"], head["rays_total"]
    kernel_s = [s_.seconds_render for s_ in head["stats_all"]]   # the render kernels alone (rehearsal included), HIP events
    seed_s = st.seconds_seed

    def pipelined(depth):
        """The same K frames with `depth` of them in flight: film k % depth on its own HIP stream, reaped just before
        its slot is reused.  The next frame's waves occupy the SIMDs the current frame's tail (a few long pixels) leaves
        idle.  Reported beside the headline, never as it."""
        pfilms = [rt.Film(W, H, device=local_rank) for _ in range(depth)]
        plist = [f.params(spp, max_depth=args.max_depth, seed=1984, variant=variant, stream=0, **knobs) for f in pfilms]
        busy = [False] * depth

        def run(n):
            for k in range(n):
                if busy[k % depth]:
                    pfilms[k % depth].finish(scene)
                pfilms[k % depth].launch(scene, plist[k % depth])
                busy[k % depth] = True
            for k in range(depth):
                if busy[k]:
                    pfilms[k].finish(scene)
                    busy[k] = False

        run(max(args.warmup, depth))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        return time.perf_counter() - t1

    pipe_elapsed = pipelined(args.pipeline) if (world == 1 and args.pipeline > 1) else None

    def emulate(n_ranks, **override):
        """Rank r of an n_ranks-way stripe split of the BASELINE frame, one rank after another on this one GPU."""
        per_rank, ppw = [], []
        kn = dict(knobs, **override)
        for r in range(n_ranks):
            f = rt.Film(W, H0, device=local_rank, stripe_rows=8, rank=r, world_size=n_ranks)
            f.render(scene, spp, max_depth=args.max_depth, variant=variant, **kn)   # warm-up
            ts = []
            for _ in range(3):
                s_ = f.render(scene, spp, max_depth=args.max_depth, variant=variant, **kn)
                ts.append(s_.seconds_seed + s_.seconds_render)
            per_rank.append(float(np.min(ts)) * 1e3)   # best of three: a rank's frame is short, the first one after a switch of films often slow
            ppw.append(int(s_.pixels_per_wave))
            del f
        return per_rank, ppw

    if rank == 0:
        def leg_value(l):
            return W * l["height"] * spp * args.steps / l["elapsed"] * 1e-6

        samples_per_step = W * H * spp
        value = leg_value(head)
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
                       "vgprs": st.kernel_vgprs, "lds_bytes": st.lds_bytes, "kind": st.kernel_kind,
                       "pixels_per_wave": st.pixels_per_wave},
        }
        if world > 1:
            out["per_rank_render_ms"] = head["per_rank_render_ms"]
            out["gather_ms"] = head["gather_ms"]
            if other is not None:
                name = "strong" if args.scaling == "weak" else "weak"
                out[name] = {
                    "scaling": name, "frame": f"{W}x{other['height']}x{spp}spp", "value": leg_value(other), "unit": "Msamples/s",
                    "ms_per_step": other["elapsed"] / args.steps * 1e3, "per_rank_render_ms": other["per_rank_render_ms"],
                    "gather_ms": other["gather_ms"], "steps": args.steps,
                    "note": ("the BASELINE frame itself striped over the ranks (north_star configs 4-5): bounded below by the longest "
                             "pixel's chain of rays, one sequential RNG stream per pixel; efficiency = value / (N x the 1-GPU value "
                             "of the same workload)") if name == "strong" else "N x the rows of the base frame, same view"}
        if pipe_elapsed is not None:
            out["pipelined"] = {"frames_in_flight": args.pipeline, "value": samples_per_step * args.steps / pipe_elapsed * 1e-6,
                                "unit": "Msamples/s", "ms_per_step": pipe_elapsed / args.steps * 1e3,
                                "note": "same K frames, launched on per-film HIP streams so that frame k+1 fills the SIMDs "
                                        "frame k's tail leaves idle; a throughput mode for frame sequences, not the headline"}
        if world == 1 and args.emulate_ranks > 1:
            per_rank, ppw = emulate(args.emulate_ranks)
            t1_ms = kernel_avg * 1e3 + seed_s * 1e3
            out["emulated_scaling"] = {
                "ranks": args.emulate_ranks, "scaling": "strong", "frame": f"{W}x{H0}x{spp}spp", "t1_ms": t1_ms,
                "per_rank_ms": per_rank, "max_rank_ms": max(per_rank), "pixels_per_wave": ppw,
                "speedup": t1_ms / max(per_rank), "efficiency": t1_ms / max(per_rank) / args.emulate_ranks,
                "note": "rank r's stripes rendered alone on this one GPU (seed + render kernels, HIP events); an N-GPU run "
                        "finishes with its slowest rank plus one gather of a few MB"}
            for v in [int(x) for x in args.emulate_ppw.split(",") if x.strip()]:
                pr, _ = emulate(args.emulate_ranks, pixels_per_wave=v)
                out.setdefault("emulated_scaling_by_pixels_per_wave", {})[str(v)] = {
                    "per_rank_ms": pr, "max_rank_ms": max(pr), "speedup": t1_ms / max(pr), "efficiency": t1_ms / max(pr) / args.emulate_ranks}
        if world == 1 and not args.no_cpu_baseline:
            base, stats, want, rows = oracle_band(wl, spp, args.max_depth, earth, args.oracle_budget)
            out["cpu_baseline"] = base
            # ---- parity: the rows of the frame just timed against the oracle at the full spp ----
            frame = head["frame"]
            par = {"variant": args.variant, "rows": [rows[0], rows[1] - 1], "width": W, "spp": spp, "tolerance": TOL,
                   "checker": "oracle/rtow_oracle.c (CPU restatement; parity unpinned, see DESIGN.md section 2)"}
            par.update(parity_of(frame[rows[0]:rows[1]], want))
            # the other build on the same rows: each 8-row stripe rendered on its own through the stripe partition
            other_variant = 1 - variant
            n_stripes = (H + 7) // 8
            got_other = np.zeros_like(want)
            for k in range(rows[0] // 8, (rows[1] + 7) // 8):
                f2 = rt.Film(W, H, device=local_rank, stripe_rows=8, rank=k, world_size=n_stripes)
                f2.render(scene, spp, max_depth=args.max_depth, variant=other_variant)
                band = f2.download()[k * 8:min(H, k * 8 + 8)]
                lo, hi = max(rows[0], k * 8), min(rows[1], k * 8 + 8)
                got_other[lo - rows[0]:hi - rows[0]] = band[lo - k * 8:hi - k * 8]
            for key, val in parity_of(got_other, want).items():   # flat: "other_variant_within_1e-5", ...
                par["other_variant_" + key] = val
            par["other_variant"] = "strict" if other_variant == 0 else "fast"
            out["parity"] = par
            # ---- roofline: fp64 VALU issue slots (the limiter); the contract's HBM algorithmic figure beside it ----
            out["roofline"] = roofline_of(args.workload, world_kind, spp, args.variant, args.flags, stats, float(st.rays), kernel_avg, int(st.kernel_kind))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
