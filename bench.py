#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp-per-step S] [--scene NAME]

Headline workload (--scene dragonglass, the default): the reference's "DragonGlass" scene at
1920x1080, max 8 bounces.  The reference's Dragon.glb is not available, so the DragonGlass material
recipe (PathTracingApplication.cpp:693-714) is applied to bunny.glb, midpoint-subdivided once to
576 k triangles (the order of the missing dragon mesh), + Floor.glb; the HDRI is the analytic sky of
pathtracer_amd.scenes (all reference HDRIs but a 1x1 black one are missing).  Data: synthetic.
Other scenes: `interior` (closed textured room lit through a window and by an emissive panel: the
stand-in for the Fireplace scene), `spheres` (sphere field, 65 materials, 16 bounces), `c1`
(BunnyClearcoat 512x512, 3 bounces: the reference's default scene at its default size).

A step is one pt_dispatch of S samples per pixel over the whole frame (S consecutive FrameCount
values of this rank's shard; the default 4 steps x 64 spp make up the config's 256 spp).  Ranks shard
by sample index (rank r renders FrameCount r+1, r+1+N, ...); the scene is replicated; per-GPU work is
fixed as N grows (weak scaling); for N > 1 the K steps are one render and end, inside the timed
region, in the one RCCL sum-reduce of the three accumulation buffers.  value = rays traced by all
ranks / max-over-ranks wall time, rays = closest-hit + shadow BVH queries counted by the kernel.

`--gpus N` without a torchrun environment starts the N ranks itself (one process per GPU).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
L2_PEAK_GBS = 34500.0   # aggregate of the eight per-XCD L2s, measured
MALL_GATHER_GBS = 8600.0  # measured ceiling for random rows served by the Infinity Cache
# The reference's Dragon.glb is missing; the Stanford dragon it is commonly built from has ~871 k
# triangles.  The bunny (144 k) subdivided once (576 k) is the closest stand-in at hand.
DEFAULT_SUBDIVIDE = 1
# Record sizes for algorithmic bytes per ray, as the kernels read them: one four-wide node (4 quantised AABBs
# + 4 refs: 64 B), one triangle (3 vertices padded), shading record, material record; one 36-byte record
# (radiance, albedo, normal: three floats each) per finished sample; 96 B of image traffic (three
# RGBA32F texels read + written) per pixel and launch.  S_NODE_EXACT: the same node with exact float boxes
# (two of the two-child 64-byte nodes the byte model was first stated with; what rounds 1-2 counted).
S_NODE, S_TRI, S_SHADE, S_MAT, S_SAMPLE, S_ACC = 64, 48, 64, 80, 36, 96
S_NODE_EXACT = 128

SCENES = {  # name -> (metric label, default width, height, bounces)
    "dragonglass": ("1080p DragonGlass", 1920, 1080, 8),
    "interior": ("1080p interior (Fireplace stand-in)", 1920, 1080, 8),
    "spheres": ("1080p sphere field", 1920, 1080, 16),
    "c1": ("512x512 BunnyClearcoat", 512, 512, 3),
    # the headline scene from 0.85 units in front of the glass mesh, which then fills the frame (the reference camera sees
    # mostly floor and sky): most rays start inside the mesh's boxes
    "closeup": ("1080p DragonGlass, close-up", 1920, 1080, 8),
}


def build_workload(args):
    """Scene buffers in the reference's layouts.  With --bvh device the nodes are left out: the
    unsorted soup goes to the GPU builder (install_scene)."""
    from pathtracer_amd import recipes, scenes
    from pathtracer_amd import renderer as R
    scene = getattr(args, "scene", "dragonglass")
    copies = getattr(args, "copies", 1)
    side, textures, camera = 1, [], None
    if scene == "interior":
        meshes, mats, textures, _ = recipes.load_scene("Interior")
        camera = recipes.INTERIOR_CAMERA
    elif scene == "spheres":
        meshes, mats = scenes.sphere_field()
        camera = dict(eye=(-14.0, 7.0, 0.0), look_at=(0.0, 0.0, 0.0))
    elif scene == "c1":
        meshes, mats = scenes.flatten(scenes.recipe("BunnyClearcoat"))
    elif copies > 1:
        meshes, mats, side = scenes.instanced_bunnies(copies)
    else:
        meshes, mats = scenes.flatten(scenes.recipe("DragonGlass"))
        if scene == "closeup":
            camera = dict(eye=(-0.15, 0.8, 0.85), look_at=(-0.15, 0.8, 0.0))
    subdivide = getattr(args, "subdivide", 0) or 0
    if subdivide > 0:
        v, i = meshes[0]
        meshes[0] = scenes.subdivide(v, i, subdivide)
    if side > 1:  # pull the camera back and up so the whole grid is in view
        camera = dict(eye=(-2.5 - 1.1 * side, 1.0 + 0.8 * side, 0.0), look_at=(0.0, 0.0, 0.0))
    prims = np.concatenate([R.assemble_primitives(v, i, k) for k, (v, i) in enumerate(meshes)])
    hdri = scenes.sky_hdri(1024, 512)
    cache = R.calculate_hdri_cache(hdri)
    work = dict(mats=mats, hdri=hdri, cache=cache, triangles=int(prims.shape[0]), side=side, textures=textures,
                camera=camera, scene=scene)
    if getattr(args, "bvh", "host") == "device":
        work.update(nodes=None, prims=R.align_primitives(prims), bvh_seconds=None)
    else:
        t0 = time.time()
        prims_sorted, nodes40 = R.build_bvh_sah(prims, 4, full_sweep=getattr(args, "bvh", "host") == "full-sweep")
        work.update(nodes=R.align_nodes(nodes40), prims=R.align_primitives(prims_sorted), bvh_seconds=time.time() - t0)
    return work


def install_scene(g, work):
    for slot, texels, srgb in work.get("textures", []):
        g.upload_texture(slot, texels, srgb)
    if work["nodes"] is None:
        t0 = time.time()
        g.build_bvh_device(work["prims"])
        work["bvh_seconds"] = time.time() - t0
        work["nodes"], work["prims"] = g.download_bvh()  # the tree the GPU built, for the CPU baseline
    else:
        g.upload_nodes(work["nodes"])
        g.upload_primitives(work["prims"])
    g.upload_materials(work["mats"])
    g.upload_environment(work["hdri"], work["cache"])


def scene_params(work, width, height, bounces):
    from pathtracer_amd import renderer as R
    params = R.default_params(width, height)
    params.max_bounces = bounces
    if work.get("camera"):
        R.set_camera(params, work["camera"]["eye"], work["camera"]["look_at"], width, height)
    return params


def cpu_baseline(work, params, width, height, threads, frames=None):
    """The CPU oracle (a scalar restatement of the reference shader, kind "port") on whole frames
    (FrameCount 1..n) of the same workload: n = `frames`, or as many as take about 15 s."""
    from oracle import binding as ob
    o = ob.Oracle()
    o.set_scene(work["nodes"], work["prims"], work["mats"])
    for slot, texels, srgb in work.get("textures", []):
        o.set_texture(slot, texels, srgb)
    o.set_environment(work["hdri"], work["cache"])
    if frames is None:  # calibrate on one frame, then size the sample to about 15 s of CPU work
        t0 = time.time()
        o.render(params, width, height, 1, 1, 1, threads=threads)
        t1 = time.time() - t0
        frames = int(max(1, min(256, round(15.0 / max(t1, 1e-3)))))
    t0 = time.time()
    _, _, _, _, ctr = o.render(params, width, height, 1, frames, 1, threads=threads)
    dt = time.time() - t0
    rays = ctr.rays_closest + ctr.rays_any
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": (f"{frames} spp (FrameCount 1..{frames}) of the full {width}x{height} frame, "
                       f"{rays} rays in {dt:.1f} s"),
            "spp_per_s": frames / dt, "seconds": dt,
            # the shader's own traversal (no pruning, 48-B nodes fetched one by one, 112-B primitives)
            "reference_traversal": {"node_fetches_per_ray": ctr.node_fetches / max(rays, 1),
                                    "pair_tests_per_ray": ctr.pair_tests / max(rays, 1),
                                    "triangle_tests_per_ray": ctr.triangle_tests / max(rays, 1),
                                    "reference_layout_bytes_per_ray": (48 * ctr.node_fetches + 112 * ctr.triangle_tests) / max(rays, 1),
                                    "max_stack_occupancy": int(ctr.max_stack), "stack_overflows": int(ctr.stack_overflows)}}


class StubShardedRenderer:
    """Stands in for ShardedRenderer under --stub (launcher and reduction logic on CPU, gloo): renders
    nothing, counts two rays per sample, keeps the three images as a CPU tensor of sample counts."""

    class _Stats:
        rays_closest = rays_any = samples = node_visits = triangle_tests = closest_hits = max_stack_depth = 0

    def __init__(self, width, height, rank, world):
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.images = torch.zeros(3, height, width, 4)
        self.total = self.images.clone()
        self.renderer = self
        self._last = self._Stats()

    def set_params(self, params):
        pass

    def clear(self):
        self.images.zero_()

    def render_step(self, step, spp, counted=False):
        self.images += float(spp)
        s = self._Stats()
        s.samples = self.width * self.height * spp
        s.rays_closest, s.rays_any = s.samples, s.samples
        s.node_visits, s.triangle_tests, s.closest_hits = 10 * s.samples, 2 * s.samples, s.samples // 2
        self._last = s

    def reduce(self):
        import torch.distributed as dist
        self.total.copy_(self.images)
        if self.world > 1:
            dist.all_reduce(self.total, op=dist.ReduceOp.SUM)

    def stats(self):
        return self._last

    def synchronize(self):
        pass

    def last_kernel_ms(self):
        return 1.0, 0.1

    def scene_bytes(self):
        return 0, 0, 0


def self_launch(args, argv):
    """`--gpus N` outside a torchrun environment: start N ranks (one process per GPU) as a child
    torch.distributed.run, which must happen before this process touches the GPU, and pass its exit
    code on; rank 0 of the child prints the line."""
    if not args.stub and os.environ.get("PT_BENCH_BACKEND", "nccl") == "nccl":
        have = torch.cuda.device_count()  # counting devices does not initialise the GPU
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd)


def pmc_summary(args, W, H, S):
    """Counter-derived figures of the trace kernel from the committed rocprofv3 PMC summary, valid
    only for the launch shape they were collected on."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), reverse=True):  # (a later round's first)
        try:
            t = json.load(open(path))
        except Exception:
            continue
        shape = (t.get("scene"), t.get("width"), t.get("height"), t.get("spp_per_step"), t.get("max_bounces"), t.get("subdivide"),
                 t.get("bvh"), t.get("copies"), int(t.get("rebuild", 0)))
        rebuilt = int(args.bvh != "device" and not getattr(args, "no_rebuild", False))
        if shape == (args.scene, W, H, S, args.bounces, args.subdivide, args.bvh, args.copies, rebuilt):
            return t
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--scene", choices=sorted(SCENES), default="dragonglass")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--subdivide", type=int, default=None,
                    help="midpoint-subdivide the first mesh this many times (dragonglass default 1: 576 k triangles; 0 with --copies)")
    ap.add_argument("--copies", type=int, default=1,
                    help="instanced-scene stress: this many bunnies flattened on a jittered grid (139 = 20 M triangles)")
    ap.add_argument("--bvh", choices=["host", "full-sweep", "device"], default="host",
                    help="host: the reference's sweep-SAH builder (bit-identical tree); device: GPU builder")
    ap.add_argument("--no-rebuild", action="store_true",
                    help="traverse the uploaded tree itself (pt_set_bvh_rebuild(ctx, 0)); by default the library checks it and "
                         "builds its own over the uploaded primitives on the GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame-by-frame", action="store_true",
                    help="skip the extra run of 64 one-sample dispatch calls (frame_by_frame)")
    ap.add_argument("--no-device-tree", action="store_true",
                    help="skip the extra measurement on the uploaded tree itself (profiling runs: one tree per kernel name)")
    ap.add_argument("--reduce-every-step", action="store_true", help="N > 1: all-reduce after every step, not once per render")
    ap.add_argument("--tuning", default="", help="scheduler settings for A/B runs: key=value,key=value (Renderer.set_tuning)")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # launcher test on CPU: gloo, no renderer
    args = ap.parse_args()
    label, dw, dh, db = SCENES[args.scene]
    W = args.width or dw
    H = args.height or dh
    args.bounces = args.bounces or db
    if args.subdivide is None:
        args.subdivide = DEFAULT_SUBDIVIDE if (args.scene in ("dragonglass", "closeup") and args.copies == 1) else 0
    S = args.spp_per_step

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    distributed = world > 1
    # every rank builds the same tree on the host before the timed region: give each its share of the cores
    # (OMP_NUM_THREADS is the job's CPU share where the launcher sets it; os.cpu_count() is the whole machine)
    omp = os.environ.get("OMP_NUM_THREADS", "")
    cores = int(omp) if omp.isdigit() and int(omp) > 0 else (os.cpu_count() or 1)
    host_threads = max(1, min(16, cores // world))
    os.environ.setdefault("PT_HOST_THREADS", str(host_threads))
    # PT_BENCH_BACKEND=gloo: a rehearsal of the N > 1 path on fewer GPUs than ranks (RCCL takes one rank per GPU):
    # ranks share the devices and the collectives go through gloo.  Not a measurement.
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    device_index = local_rank
    if not args.stub:
        if backend != "nccl":
            device_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(device_index)
    over_gloo = args.stub or backend != "nccl"
    if distributed:
        import datetime
        import faulthandler
        import torch.distributed as dist
        # A rank that blocks (rendezvous, a collective whose peer died) must end with a traceback, not sit
        # silently until somebody's outer limit: PT_BENCH_WATCHDOG_S seconds after start every rank dumps its
        # threads' stacks to stderr and exits (0 = off; the default leaves a real N-GPU run ample time).
        watchdog = float(os.environ.get("PT_BENCH_WATCHDOG_S", "120" if over_gloo else "1500"))
        if watchdog > 0:
            faulthandler.dump_traceback_later(watchdog, exit=True)
        if over_gloo:
            # gloo picks its interface from the host name unless told otherwise; MASTER_ADDR only places the store
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=600))

    def sync():
        if not args.stub:
            torch.cuda.synchronize()

    if args.stub:
        work = dict(triangles=0, bvh_seconds=0.0, side=1)
        sr = StubShardedRenderer(W, H, rank, world)
        g, params = sr, None
    else:
        from pathtracer_amd.distributed import ShardedRenderer
        work = build_workload(args)
        sr = ShardedRenderer(W, H, device_index, rank, world)
        g = sr.renderer
        if args.tuning:  # before the scene goes in: some settings steer the device BVH builder
            g.set_tuning(**{k: int(v) for k, v in (item.split("=") for item in args.tuning.split(","))})
        if args.no_rebuild:
            g.set_bvh_rebuild(False)
        install_scene(g, work)
        params = scene_params(work, W, H, args.bounces)
        sr.set_params(params)

    # traversal statistics of one step (untimed, counting build of the same kernel, same frames)
    sr.clear()
    sr.render_step(0, S, counted=True)
    g.synchronize()
    st = g.stats()
    rays_step = st.rays_closest + st.rays_any
    bytes_per_ray = (st.node_visits * S_NODE + st.triangle_tests * S_TRI + st.closest_hits * (S_SHADE + S_MAT)) / rays_step
    bytes_per_ray_exact_nodes = bytes_per_ray + st.node_visits * (S_NODE_EXACT - S_NODE) / rays_step
    # the trace kernel writes one 36-byte record per finished sample; the resolve kernel reads the
    # records back and updates the three images once per launch (16 B read + 16 B written each)
    bytes_per_launch = bytes_per_ray * rays_step + st.samples * S_SAMPLE
    resolve_bytes = st.samples * S_SAMPLE + W * H * S_ACC

    sr.clear()
    for w in range(args.warmup):
        sr.render_step(w, S)
        sr.reduce()
    if distributed and args.warmup == 0:
        sr.reduce()  # untimed: the first collective of a communicator sets its channels up
    sync()
    if distributed:
        dist.barrier()
    sync()
    rays = 0
    kernel_ms = []
    step_ms = []  # this rank's wall time per step (each ends in the counter read, which synchronises)
    t0 = time.perf_counter()
    for k in range(args.steps):
        t_step = time.perf_counter()
        sr.render_step(k, S)
        if args.reduce_every_step:
            sr.reduce()
        # counters and event times are read after the stream work of this step was queued; reading
        # them synchronises this rank's stream (part of the measured time, as a real caller would)
        s = g.stats()
        rays += s.rays_closest + s.rays_any
        try:
            kernel_ms.append(g.last_kernel_ms())
        except Exception:  # the wavefront pipeline (--tuning pipeline=1) has no single trace kernel: whole dispatch
            kernel_ms.append((g.last_dispatch_ms(), 0.0))
        step_ms.append((time.perf_counter() - t_step) * 1e3)
    if not args.reduce_every_step:
        sr.reduce()  # one collective per render (the K steps)
    sync()
    if distributed:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0

    rank_ms = [elapsed * 1e3]
    if distributed:
        dev = "cpu" if over_gloo else "cuda"  # scalars: host tensors under gloo, device tensors under RCCL
        t = torch.tensor([elapsed, float(rays)], device=dev, dtype=torch.float64)
        tmax, tmin = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_ms = [float(tmin[0]) * 1e3, float(tmax[0]) * 1e3]
        elapsed, rays = float(tmax[0]), float(t[1])

    if rank == 0:
        total_spp = args.steps * S * world
        avg_kernel_s = float(np.mean([t for t, _ in kernel_ms])) / 1e3
        avg_resolve_s = float(np.mean([r for _, r in kernel_ms])) / 1e3
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        scene_bytes = int(sum(g.scene_bytes()))
        cache_resident = scene_bytes < 256e6
        pmc = None if args.stub else pmc_summary(args, W, H, S)
        traffic = pmc.get("bytes_per_launch") if pmc else None
        if args.copies > 1:
            what = f"DragonGlass recipe on {args.copies} flattened bunnies + floor ({work['triangles']} triangles), "
            label = f"{W}x{H} DragonGlass on {work['triangles'] / 1e6:.0f} M triangles"
        elif args.scene in ("dragonglass", "closeup"):
            what = (f"DragonGlass recipe on bunny.glb (subdivided x{args.subdivide}) + Floor.glb stand-in ({work['triangles']} triangles), " +
                    ("camera 0.85 units in front of the mesh (it fills the frame), " if args.scene == "closeup" else ""))
        elif args.scene == "interior":
            what = (f"interior stand-in for the Fireplace scene: closed textured room with one window, AreaLight and Mill content "
                    f"({work['triangles']} triangles), ")
        elif args.scene == "spheres":
            what = f"sphere field, 64 spheres with 65 seeded materials ({work['triangles']} triangles), "
        else:
            what = f"BunnyClearcoat recipe on bunny.glb + Floor.glb ({work['triangles']} triangles), "
        if (W, H) != (dw, dh) and args.copies == 1:
            label = label.replace("1080p", f"{W}x{H}").replace("512x512", f"{W}x{H}")
        out = {
            "metric": f"Mrays/s at {label} (closest-hit + shadow BVH queries per second)",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median": float(np.median(step_ms)),
            "ms_per_step_min": float(np.min(step_ms)),
            "ms_per_step_max": float(np.max(step_ms)),
            "rank_ms_min": rank_ms[0],
            "rank_ms_max": rank_ms[-1],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "spp_per_s": total_spp / elapsed,
            "msamples_per_s": total_spp * W * H / elapsed / 1e6,
            "config": {
                "workload": ("stub renderer (launcher test)" if args.stub else
                             what + f"{W}x{H}, max {args.bounces} bounces, {S} spp per step per GPU, analytic sky HDRI 1024x512, "
                             "BVH " + {"device": "built by the GPU clustering builder from the triangle soup",
                                       "host": "uploaded from the sweep-SAH host builder (the reference's tree)",
                                       "full-sweep": "uploaded from the sweep-SAH host builder without the reference's cost cap"}[args.bvh] +
                             ("" if args.bvh == "device" else
                              ", traversed as uploaded (rebuild off)" if args.no_rebuild else
                              "; the library checks the uploaded tree and traverses its own, built on the GPU over the uploaded "
                              "primitives (its default, pt_set_bvh_rebuild)")),
                "scene": args.scene, "width": W, "height": H, "max_bounces": args.bounces, "spp_per_step_per_gpu": S,
                "total_spp": total_spp, "parallelism": f"sample-sharded x{world}" if world > 1 else "single GPU",
                "reduce": ("none (one GPU)" if world == 1 else
                           "REHEARSAL over gloo with ranks sharing GPUs: not a measurement" if backend != "nccl" and not args.stub else
                           "one RCCL all-reduce per step" if args.reduce_every_step else "one RCCL all-reduce per render (after the last step, timed)"),
                "bvh_build_seconds": round(work["bvh_seconds"], 3),
                "tree_traversed": "device-built" if (args.bvh == "device" or not args.no_rebuild) else "as uploaded",
            },
            "roofline": {
                # `achieved` = ALGORITHMIC bytes per launch over the trace kernel's HIP-event time, measured in this run.
                # `peak` / `frac` rate them against the memory level that serves them: the eight L2s for a scene the
                # caches hold, HBM for one they cannot; `frac_hbm` is the same figure against the HBM peak.
                "bound": "hbm",  # a memory-system roofline: the path has no dense contraction, no MFMA
                "limited_by": pmc.get("limited_by") if pmc else None,
                "achieved": achieved,
                "peak": L2_PEAK_GBS if cache_resident else HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / (L2_PEAK_GBS if cache_resident else HBM_PEAK_GBS),
                "frac_hbm": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "level": ("L2 (aggregate of the eight per-XCD L2s; the scene is cache resident)" if cache_resident
                          else "HBM (scene exceeds the 256 MB Infinity Cache)"),
                "note": ("measured in this run: achieved, frac, frac_hbm, kernel_ms_avg, the per-ray counts, resolve_kernel, ms_per_step_*. "
                         "Read from pmc_source (rocprofv3 counter passes of the same command on the same launch shape, not this run): "
                         "traffic, fabric_gbps, fabric_frac, l1_busy, l1_stall, valu_pipe_share, valu_busy, lane_utilisation, "
                         "waves_per_simd, l2_hit, l2_request_gbps, l1_l2_read_latency_cycles, limited_by. achieved counts ALGORITHMIC bytes: on a cache-resident scene the vector L1s "
                         "and the L2s serve them, so frac_hbm can exceed 1 and peak is the L2s' measured aggregate; traffic / "
                         "fabric_gbps are the bytes leaving the L2s; l1_busy / l1_stall: share of the CUs' cycles in which the vector "
                         "L1 takes an access, one per cycle, or is held up by misses in flight (the two always add up to 1); lane_utilisation counts a "
                         "lane as active while it runs the half of the unbranched traversal step it has no use for (on record 0, results masked)"),
                "frac_l2": achieved / L2_PEAK_GBS,
                "fabric_gbps": (traffic / avg_kernel_s / 1e9) if traffic else None,
                "fabric_frac": ((traffic / avg_kernel_s / 1e9) / (MALL_GATHER_GBS if cache_resident else HBM_PEAK_GBS)) if traffic else None,
                "fabric_peak": MALL_GATHER_GBS if cache_resident else HBM_PEAK_GBS,
                "l1_busy": pmc.get("l1_busy") if pmc else None,
                "l1_stall": pmc.get("l1_stall") if pmc else None,
                "valu_pipe_share": pmc.get("valu_pipe_share") if pmc else None,
                "valu_busy": (pmc.get("valu_busy_raw", pmc.get("valu_busy"))) if pmc else None,
                "lane_utilisation": pmc.get("lane_utilisation") if pmc else None,
                "waves_per_simd": pmc.get("waves_per_simd") if pmc else None,
                "l2_hit": pmc.get("l2_hit") if pmc else None,
                "l2_request_gbps": pmc.get("l2_request_gbps") if pmc else None,  # L2 requests x 128-byte line / kernel time
                "l1_l2_read_latency_cycles": pmc.get("l1_l2_read_latency_cycles") if pmc else None,
                "pmc_source": pmc.get("source") if pmc else None,
                "kernel": "renderKernel" if avg_resolve_s > 0 else "traverseKernel + shadeKernel (whole dispatch)",
                "kernel_ms_avg": avg_kernel_s * 1e3,
                "algorithmic_bytes_per_ray": bytes_per_ray,
                "bytes_per_ray_with_exact_128B_nodes": bytes_per_ray_exact_nodes,  # the earlier rounds' accounting, for comparison
                "rays_per_launch": rays_step,
                "node_visits_per_ray": st.node_visits / rays_step,
                "triangle_tests_per_ray": st.triangle_tests / rays_step,
                "max_stack_depth": st.max_stack_depth,
                "scene_bytes": scene_bytes,
                "resolve_kernel": ({"kernel_ms_avg": avg_resolve_s * 1e3, "algorithmic_bytes": resolve_bytes,
                                    "achieved": resolve_bytes / avg_resolve_s / 1e9, "unit": "GB/s", "bound": "hbm",
                                    "frac": resolve_bytes / avg_resolve_s / 1e9 / HBM_PEAK_GBS} if avg_resolve_s > 0 else None),
            },
        }
        if world == 1 and not args.stub and not args.no_frame_by_frame:
            # Outside the timed region, for the record: the reference viewer's call pattern, one sample per
            # pixel per dispatch call, on the library's own stream (where consecutive calls overlap).
            try:
                g.set_stream(0)
                frames = 64
                g.dispatch(1, 1, 1)
                g.synchronize()
                t1 = time.perf_counter()
                for f in range(frames):
                    g.dispatch(2 + f, 1, 1)
                g.synchronize()
                dt = (time.perf_counter() - t1) / frames
                st1 = g.stats()
                out["frame_by_frame"] = {"ms_per_frame": dt * 1e3, "value": (st1.rays_closest + st1.rays_any) / dt / 1e6,
                                         "unit": "Mrays/s", "frames": frames, "spp_per_dispatch": 1}
            except Exception as e:
                out["frame_by_frame"] = {"error": str(e)}
            finally:
                g.set_stream(sr.stream.cuda_stream)
        if world == 1 and args.bvh == "host" and args.copies == 1 and not args.no_device_tree and not args.no_rebuild and not args.stub:
            # Outside the timed region, for the record: the same steps with the uploaded tree itself traversed
            # (pt_set_bvh_rebuild(ctx, 0)): the reference's builder caps its split cost, see --bvh.
            try:
                g.set_bvh_rebuild(False)
                g.upload_nodes(work["nodes"])
                g.upload_primitives(work["prims"])
                sr.clear()
                sr.render_step(0, S)
                g.synchronize()
                alt_rays, t1 = 0, time.perf_counter()
                for k in range(args.steps):
                    sr.render_step(k, S)
                    st2 = g.stats()
                    alt_rays += st2.rays_closest + st2.rays_any
                g.synchronize()
                dt = time.perf_counter() - t1
                out["uploaded_tree_traversed"] = {"value": alt_rays / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt / args.steps * 1e3,
                                                  "note": "pt_set_bvh_rebuild(ctx, 0): the reference builder's tree as uploaded"}
            except Exception as e:  # never lose the benchmark line over the extra measurement
                out["uploaded_tree_traversed"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1 and not args.stub:
            threads = min(16, os.cpu_count() or 1)
            try:
                # c1 is the configuration the CPU baseline is defined on: all of its 64 spp (about 40 M rays)
                out["cpu_baseline"] = cpu_baseline(work, params, W, H, threads, frames=64 if args.scene == "c1" else None)
            except Exception as e:  # the oracle library is a test artefact: report, do not lose the line
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
