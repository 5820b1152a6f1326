#!/usr/bin/env python3
"""Benchmark of the octree-intersection + path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full frame of the workload: every rank renders its slice of the samples of
every pixel into a full-frame fp64 radiance buffer (device resident), the buffers are summed
onto rank 0 (RCCL) and rank 0 resolves the frame to bytes.  At N = 1 the workload is the
single-GPU benchmark configuration (bunny-class ~69 k triangle mesh in a triangle Cornell box,
octree leaf capacity 128 as in the reference, 1024 x 1024, 256 spp).  For N > 1 the per-GPU
work is kept fixed by default (weak scaling): the frame has 256*N samples per pixel, rank g
renders sample indices [256 g, 256 (g+1)).  `--scaling strong` keeps the frame fixed instead
(bunny: 256 spp, atrium: 1024 spp = the target configuration) and deals its samples to the ranks,
so that the times of an N = 1 and an N = 8 run give the efficiency T1 / (N TN) directly.

The timed region starts with the scene already uploaded (octree build + upload are one-time
costs outside it) and contains render + reduce + resolve.  `value` = all rays traced by all
ranks (camera + bounce + shadow rays, counted on the device) / seconds, in Mrays/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raytracer_amd.parallel import dist_env, sample_slice  # noqa: E402  (no GPU library behind this import)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
VALU_FP64_PEAK = 39.3e12       # 256 CU x 4 SIMD x 16 fp64 lanes/clk x 2.4 GHz (operations/s, a fused multiply-add = 1:
                               # the path is built without contraction, so an operation is one flop)
RAY_RECORD_BYTES = 88.0        # a pending bounce ray: 11 eight-byte fields
SHADOW_RECORD_BYTES = 84.0     # a pending shadow ray: 10 eight-byte fields and the pixel index
SHADE_READ_BYTES = 40.0        # what shading reads of every ray: meta word, hit id, throughput
BOX_TEST_OPS = 23.0            # one single-precision box pre-test: 6 conversions, 6 multiply-adds, 10 minima / maxima, 1 compare


def shade_bytes(flow):
    """Bytes the shading kernel has to move for the rays it handled (a lower bound: the 64 bytes
    more that a shaded hit reads and the 24 bytes a contributing ray adds to its pixel are left
    out): the fields every ray is read for, and one record written per follow-up ray."""
    return (SHADE_READ_BYTES * flow["shaded_in"] + RAY_RECORD_BYTES * (flow["bounce_class0"] + flow["bounce_class1"])
            + SHADOW_RECORD_BYTES * (flow["shadow_class0"] + flow["shadow_class1"]))


def cpu_baseline(desc, cam, W, H, n, m, seed, budget_s=20.0):
    """The CPU oracle (a port of the reference path) on this box's host cores, on a bounded
    sample of the same workload: whole image rows spread over the frame, all samples."""
    from oracle import orc
    # every host core this process may use (the GPU box sets OMP_NUM_THREADS to its share of the machine)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    so = orc.OracleScene(desc)
    c = orc.camera_init(**cam)
    prm = render_params(W, H, n, m, seed=seed)
    # calibrate on one row, then take as many evenly spaced rows as fit the budget
    t0 = time.time()
    _, k = so.render(c, prm, threads=threads, pix_range=(W * (H // 2), W * (H // 2) + W))
    dt = time.time() - t0
    rows = int(max(1, min(H, budget_s / max(dt, 1e-3))))
    ys = np.unique(np.linspace(0, H - 1, rows).astype(int))
    rays = 0
    t0 = time.time()
    for y in ys:
        _, k = so.render(c, prm, threads=threads, pix_range=(W * int(y), W * int(y) + W))
        rays += int(k[0])
    dt = time.time() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{len(ys)} of {H} image rows (evenly spaced, full {n * m} spp each): {rays} rays in {dt:.1f} s"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="image width = height")
    ap.add_argument("--spp-root", type=int, default=16, help="n of the n x n jitter table per GPU")
    ap.add_argument("--workload", choices=["bunny", "atrium"], default="bunny",
                    help="bunny: the single-GPU benchmark configuration (default); atrium: the Sponza-class scene "
                         "(~262 k triangles, rectangular light) at 1920 x 1080, 128 spp per GPU unless --spp-root is given")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: samples per GPU fixed (default); strong: the frame fixed (bunny 256 spp, atrium 1024 spp), "
                         "its samples dealt to the ranks")
    ap.add_argument("--reduce", choices=["torch", "native"], default="torch",
                    help="the frame's one exchange step: torch.distributed.reduce (RCCL through torch; default) or the library's own "
                         "rt_reduce_and_resolve (ncclReduce through the RCCL the process holds, unique id broadcast by torch)")
    ap.add_argument("--workspace-gb", type=float, default=0.0,
                    help="upper bound on the pending-ray stacks in GB (0 = the library's default: 16 Mi rays a round, at most a "
                         "quarter of the free device memory); RT_WAVEFRONT_RAYS in the environment sets the rays per round")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    return ap.parse_args(argv)


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: this process starts N rank processes (one per
    GPU, the environment torch.distributed.run would give them) and never touches a GPU itself --
    nothing that initialises the HIP runtime has been imported at this point.  Rank 0's JSON line is
    passed through as this process's own; any rank failing fails the run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RT_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    for p in procs[1:]:
        try:
            codes.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:       # a rank left behind by a failed peer
            p.kill()
            codes.append(p.wait())
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit(f"bench.py: rank processes failed (rank, exit code): {bad}")


def main():
    a = parse_args()
    rank, local_rank, world = dist_env()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if world == 1 and a.gpus > 1:
        return spawn_ranks(a)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started {world} ranks")

    global np, torch, dist, hip, scenes, render_params
    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracer_amd import hip, scenes
    from raytracer_amd.desc import render_params

    # one rank per GPU; RT_BENCH_BACKEND=gloo lets several ranks share a card to rehearse the N > 1
    # path on a one-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"--gpus {world} needs {world} devices, this machine shows {ndev} "
                         "(RT_BENCH_BACKEND=gloo rehearses the N > 1 path with the ranks sharing a card)")
    if backend != "nccl":
        local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    hip.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    native = world > 1 and a.reduce == "native"
    comm = None
    if native:
        if backend != "nccl":
            raise SystemExit("--reduce native needs one device per rank (RCCL)")
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(hip.Comm.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        comm = hip.Comm(unique_id=bytes(uid.cpu().numpy().tobytes()), nranks=world, rank=rank, device=local_rank)

    # rays per round: the library's default is 16 Mi; the benchmark has the card to itself and takes 32 Mi
    # (+3 % rays/s for twice the stack memory, reported as workspace_gb)
    os.environ.setdefault("RT_WAVEFRONT_RAYS", str(32 << 20))
    seed = 1243                                 # the reference's srand value (scene.cpp:22)
    if a.workload == "atrium":
        W, H = (1920, 1080) if a.size == 1024 else (a.size * 16 // 9, a.size)
        n = a.spp_root
        m = (8 if a.spp_root == 16 else a.spp_root) * world      # 128 spp per GPU: 1024 spp on 8 GPUs
        if a.scaling == "strong":
            n, m = (32, 32) if a.spp_root == 16 else (a.spp_root, a.spp_root)   # the target frame: 1024 spp whatever N is
        desc, cam = scenes.atrium_scene()
        label = "Sponza-class atrium"
    else:
        W = H = a.size
        n = a.spp_root
        m = a.spp_root * (world if a.scaling == "weak" else 1)   # weak scaling: 256 samples per pixel per GPU
        desc, cam = scenes.bunny_cornell_scene()    # 69 564-triangle mesh, Cornell box of triangle meshes, sphere light
        label = "bunny-class mesh"
    per_gpu = -(-n * m // world)
    ntris = sum(len(x) for x in desc.meshes)
    t0 = time.time()
    sc = hip.HipScene(desc)
    if a.workspace_gb > 0:
        sc.set_workspace_limit(int(a.workspace_gb * 1e9))
    upload_s = time.time() - t0
    c = hip.camera_init(**cam)
    b, e = sample_slice(rank, world, n * m)
    prm = render_params(W, H, n, m, sample_begin=b, sample_end=e, seed=seed)

    img = torch.zeros(H, W, 3, dtype=torch.float64, device=dev)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    bgr = torch.empty(H, W, 3, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def exchange():
        """The frame's one exchange step and the resolve on rank 0."""
        if native:
            comm.reduce_and_resolve(0, img.data_ptr(), W, H, n * m, bgr.data_ptr() if rank == 0 else 0, stream=stream)
            return
        if world > 1:
            dist.reduce(img, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            hip.resolve_bgr8_dev(img.data_ptr(), W, H, n * m, bgr.data_ptr(), stream)

    def step():
        img.zero_()
        sc.render_dev(c, prm, img.data_ptr(), cnt.data_ptr(), stream)
        exchange()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # set up the collective's connections outside the timed region whatever --warmup is
        # (the first reduce of a communicator establishes them)
        exchange()
    for _ in range(a.warmup):
        step()
    fence()
    cnt.zero_()
    sc.set_profiling(True)
    sc.get_profile(reset=True)
    sc.get_kernel_stats(reset=True)
    sc.get_ray_flow(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    sc.check()
    prof = sc.get_profile()
    kst = sc.get_kernel_stats()
    flow = sc.get_ray_flow()
    sc.set_profiling(False)

    small = dev if backend == "nccl" else torch.device("cpu")      # where the few bytes of bookkeeping are exchanged
    tmax = torch.tensor([dt], dtype=torch.float64, device=small)
    total = cnt.to(small)
    rank_ms = [dt / max(a.steps, 1) * 1e3]
    if world > 1:
        every = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(every, tmax)
        rank_ms = [float(x.item()) / max(a.steps, 1) * 1e3 for x in every]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total = total.cpu().numpy().astype(np.float64)

    if rank == 0:
        steps = max(a.steps, 1)
        rays_per_step = total[0] / steps
        paths_per_step = float(W) * H * n * m
        # Dominant kernel: rt::k_trace, the octree engine over bounce rays, timed with HIP events around
        # every launch on the stream it runs on; the other stages of the closest-hit search are listed
        # beside it, each with its own device counters.  The scene data stays in the caches, so what binds
        # these kernels is vector arithmetic, not HBM.  `frac` prices what was executed for the launch's
        # rays (slab tests, fp64 primitive tests and box pre-tests made, counted on the device) against
        # the fp64 vector peak; `algorithmic_frac` prices the reference's walk per unit (N_n, N_l, N_p)
        # whether or not the box pre-tests let the kernel skip a primitive, and may exceed 1.  HBM traffic
        # comes from counter passes of the same command (profiles/, replayed here: a --pmc pass cannot
        # run inside this process).
        pmc = {}
        pfile = os.path.join(ROOT, "profiles", "pmc_" + a.workload + ".json")
        # the counter passes were made on the default frame of the workload (per GPU: weak scaling keeps it);
        # any other frame has no measured traffic to report
        if os.path.exists(pfile) and a.size == 1024 and a.spp_root == 16 and (a.scaling == "weak" or world == 1):
            try:
                pmc = json.load(open(pfile))
            except Exception:
                pmc = {}
        src = f"replayed from profiles/pmc_{a.workload}.json (rocprofv3 --pmc passes of this command, {pmc.get('collected', 'n/a')})" if pmc else None

        def per_launch(fig, launches_per_step):
            """HBM bytes per launch: the profiled step's total for the kernel (the same frame) over this run's launches per step."""
            if "fetch_bytes_total_corrected" not in fig:
                return None
            return (fig["fetch_bytes_total_corrected"] + fig["write_bytes_total"]) / max(launches_per_step, 1.0)

        def trace_entry(name, st, fig):
            launches = max(st["launches"], 1.0)
            avg_ms = st["ms"] / launches
            alg = 26.0 * 9.0 * st["inner"] + 26.0 * st["leaf"] + 50.0 * st["prim"]
            exe = 26.0 * 9.0 * st["inner"] + 26.0 * st["leaf"] + 50.0 * st["fp64_tests"] + BOX_TEST_OPS * st["box_tests"]
            rate = lambda ops: ops / (st["ms"] * 1e-3) if st["ms"] > 0 else 0.0
            rays = max(st["rays"], 1.0)
            issue, lanes = fig.get("valu_issue_share"), fig.get("active_lanes_per_valu_inst")
            return {"kernel": name, "bound": "valu-fp64",
                    "achieved": rate(exe) / 1e12, "peak": VALU_FP64_PEAK / 1e12, "unit": "TFLOP/s", "frac": rate(exe) / VALU_FP64_PEAK,
                    "ops_model": f"operations executed for the launch's rays, from device counters: 26 per slab test (9 at an inner node, 1 at a leaf) + 50 per fp64 "
                                 f"primitive test made + {BOX_TEST_OPS:.0f} per single-precision box test made (node bookkeeping, sorting and addressing are not counted)",
                    "executed_ops_per_launch": exe / launches,
                    "traffic": per_launch(fig, st["launches"] / steps), "traffic_source": src if fig else None,
                    "launches": st["launches"], "avg_launch_ms": avg_ms,
                    "algorithmic_ops_per_launch": alg / launches, "algorithmic_achieved": rate(alg) / 1e12, "algorithmic_frac": rate(alg) / VALU_FP64_PEAK,
                    "algorithmic_model": "the reference's walk priced per unit: 26 per slab test + 50 per primitive it tests (N_p), whether or not the box pre-tests let "
                                         "the kernel skip that primitive -- a rate of useful work (the figure this benchmark called frac before the pre-tests existed), not bounded by 1",
                    "rays": st["rays"], "inner_visits_per_ray": st["inner"] / rays, "leaf_visits_per_ray": st["leaf"] / rays,
                    "covered_prims_per_ray": st["prim"] / rays, "executed_fp64_tests_per_ray": st["fp64_tests"] / rays,
                    "executed_box_tests_per_ray": st["box_tests"] / rays,
                    "hw_valu_issue_share": issue, "hw_active_lanes_per_valu_inst": lanes,
                    "hw_valu_lane_util": issue * lanes / 64.0 if issue is not None and lanes is not None else None,
                    "hw_lanes_note": "lanes enabled by EXEC, not lanes whose results are kept: the engine's node step runs without "
                                     "per-lane branches (every lane computes, outcomes are selected), so lanes that stand at no node count as active",
                    "share_of_step_time": st["ms"] * 1e-3 / dt}

        pm_names = ("k_trace", "k_trace_shadow", "k_sweep", "k_sweep_shadow")
        full = ("rt::k_trace (octree engine, bounce rays)", "rt::k_trace_shadow (octree engine, shadow rays)",
                "rt::k_sweep + rt::k_trace_retry (bounce rays outside the octrees)", "rt::k_sweep_shadow + rt::k_trace_shadow_retry (shadow rays outside the octrees)")
        entries = [trace_entry(full[i], kst[hip.HipScene.KERNEL_STAGES[i]], pmc.get(pm_names[i], {})) for i in range(4)]
        shade_launches = max(prof["shade_launches"], 1.0)
        shade_avg_ms = prof["shade_ms"] / shade_launches
        shade_b = shade_bytes(flow) / shade_launches
        shade_gbs = shade_b / (shade_avg_ms * 1e-3) / 1e9 if shade_avg_ms > 0 else 0.0
        pshade = pmc.get("shade", {})
        out = {
            "metric": "Mrays/s (whole job; spp/s and roofline fractions alongside)",
            "value": rays_per_step * steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"{label} ({ntris} triangles) in a triangle Cornell box via octree (leaf capacity 128), " if a.workload == "bunny"
                                    else f"{label} ({ntris} triangles, rectangular light) via octrees (leaf capacity 128), ")
                                   + f"{W}x{H} @ {per_gpu} spp per GPU ({n * m} spp total), max depth 40 / roulette from 12",
                       "width": W, "height": H, "spp_total": n * m, "spp_per_gpu": per_gpu, "triangles": ntris,
                       "parallelism": f"spp-partition x{world}, sum-reduce to rank 0"},
            "rank_ms_per_step": rank_ms,
            "rccl_nranks": (comm.size() if native else dist.get_world_size()) if world > 1 and backend == "nccl" else (1 if world == 1 else 0),
            "reduce": "none (one rank)" if world == 1 else ("rt_reduce_and_resolve (ncclReduce)" if native else f"torch.distributed.reduce ({backend})"),
            "launcher": "bench.py's own rank processes" if os.environ.get("RT_BENCH_SPAWNED") else ("torch.distributed.run" if world > 1 else "none"),
            "spp_per_s": n * m * steps / dt,
            "mpaths_per_s": paths_per_step * steps / dt / 1e6,
            "rays_per_path": rays_per_step / paths_per_step,
            "upload_s": upload_s,
            "workspace_gb": sc.workspace_bytes() / 1e9,
            "roofline": entries[0],
            "roofline_other_trace_kernels": entries[1:],
            "roofline_shade": {"kernel": "rt::k_shade (shading + ray append)", "bound": "hbm",
                               "achieved": shade_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": shade_gbs / HBM_PEAK_GBS,
                               "traffic": per_launch(pshade, prof["shade_launches"] / steps), "traffic_source": src if pshade else None,
                               "launches": shade_launches, "avg_launch_ms": shade_avg_ms, "algorithmic_bytes_per_launch": shade_b,
                               "share_of_step_time": prof["shade_ms"] * 1e-3 / dt},
            "stage_ms_per_step": {"trace": prof["trace_ms"] / steps, "shade": prof["shade_ms"] / steps,
                                  "rounds": prof["rounds"] / steps},
        }
        tr = entries[0]["traffic"]
        if tr and entries[0]["avg_launch_ms"] > 0:
            gbs = tr / (entries[0]["avg_launch_ms"] * 1e-3) / 1e9
            out["roofline_hbm"] = {"kernel": entries[0]["kernel"], "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src,
                                   "note": "measured traffic of the profiled run over this run's launch time: the scene is cache resident, HBM does not bind"}
        for key in ("roofline_shade", "roofline_hbm"):
            if key in out and not (0.0 <= out[key]["frac"] <= 1.0):
                raise SystemExit(f"bench.py: {key}.frac = {out[key]['frac']:.3f} is not a fraction of a peak -- the model behind it is wrong")
        for e in entries:
            if not (0.0 <= e["frac"] <= 1.0):
                raise SystemExit(f"bench.py: frac = {e['frac']:.3f} of {e['kernel']} is not a fraction of a peak -- the count behind it is wrong")
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(desc, cam, W, H, n, m, seed, a.cpu_budget)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
