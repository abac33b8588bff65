#!/usr/bin/env python3
"""bench.py — throughput of the ray-marching path on MI355X, BASELINE.json's metric and config.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--generator Rectilinear|Fast]

A "step" is one full frame of the headline workload (4096x2048 panorama, 3x3 synthetic DTED level-2
tiles, simulation_step 100 m, max_distance 200 km, spherical Earth + US-76 refraction) through the C ABI
(atmrt_generate_image_device): terrain tiles and parameters are resident in HBM before the timed region, the
per-pixel result planes stay in HBM.  With N > 1 (one process per GPU; `python bench.py --gpus N` starts the N ranks itself as a
torch.distributed.run child, or run it under torchrun) every rank marches the pixel-column tile the LIBRARY assigns it, and the
library itself — C++ below the C ABI, csrc/atmrt_multi.hip — all-gathers the tiles' planes (one 84 B/pixel slab per rank) with ONE
ncclAllGather over RCCL/xGMI and permutes them into the [H][W] image, inside the step.

Prints ONE JSON line (rank 0).  `value` = ray-steps marched by all ranks per second, where a ray-step is one
sample-pair evaluation of get_single_pixel's loop under the reference's termination rule (utils.rs:211-287).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_ISSUE_PEAK = 256 * 4 * 16 * 2.4e9  # lane-instructions/s
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6   # vendor-published FP64 vector peak, secondary figure


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(tiles, level):
    """The oracle (CPU restatement of the reference, glibc-libm flavour, OpenMP) on a bounded sample of the SAME scene and the
    SAME terrain tiles the GPU ran: Rectilinear on every 4th pixel in x and y (1024x512 = 1/16 of the pixels, same field of view,
    step and max_distance; ~10-20 s on 16 threads), and the reference's default generator Fast on the whole 4096x2048 frame
    (~6 s).  A reported baseline, not a target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle
    from atm_raytracer_amd import synth
    oracle = Oracle("libm")
    t = oracle.terrain_new(tiles)
    # the GPU box gives one GPU a 16-core share of the host; never fan out over the whole machine
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("ATMRT_CPU_THREADS", "16")))
    out = {}
    try:
        for generator, (w, h) in (("Rectilinear", (1024, 512)), ("Fast", (4096, 2048))):
            cfg, _ = synth.scene("headline", w, h, generator=generator, level=301)  # the tiles of the GPU run are used, not these
            t0 = time.perf_counter()
            res = oracle.generate(cfg.params, cfg.atmosphere, t, [], cores)
            dt = time.perf_counter() - t0
            out[generator] = {"value": res["ray_steps"] / dt, "unit": "ray-steps/s", "mpixels_per_s": w * h / dt / 1e6, "seconds": dt,
                              "ray_steps": res["ray_steps"], "pixels": f"{w}x{h}"}
    finally:
        oracle.terrain_free(t)
    r = out["Rectilinear"]
    return {"value": r["value"], "unit": "ray-steps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"Rectilinear generator, headline scene sampled at {r['pixels']} px = 1/16 of the 4096x2048 pixels (same fov, step, "
                      f"max_distance), the run's own DTED level-{level} tiles: {r['ray_steps']} ray-steps in {r['seconds']:.2f} s on {cores} "
                      f"OpenMP threads; oracle/liboracle_libm.so (reference algorithm incl. eager normals per sample)",
            "mpixels_per_s": r["mpixels_per_s"],
            "fast": {k: out["Fast"][k] for k in ("value", "unit", "mpixels_per_s", "seconds", "pixels")}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--generator", default="Rectilinear", choices=["Rectilinear", "Fast", "InterpolatingRectilinear"])
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=2048)
    ap.add_argument("--dted-level", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--also-fast", action="store_true", help="(default behaviour) also time the Fast generator and report it under 'fast'")
    ap.add_argument("--only", action="store_true", help="time only --generator, skip the secondary generators")
    ap.add_argument("--objects", type=int, default=0, help="BASELINE config 5: add N scene objects (70%% frusta, 30%% billboards)")
    ap.add_argument("--terrain-alpha", type=float, default=1.0)
    ap.add_argument("--scene", default="headline", choices=["headline", "S2", "S3", "S4"], help="SURVEY §8(d) scene (tiles, observer, fov)")
    ap.add_argument("--step", type=float, default=None, help="simulation_step override [m]")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: one process per GPU is the contract, so start the N ranks as a child
        # torch.distributed.run job (this process has not touched the GPU and never will), pass its JSON line through and exit with
        # its code.  Under an outer torchrun (RANK set) this branch is not taken.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"bench.py: launching {args.gpus} ranks: {' '.join(cmd)}")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import numpy as np
    import torch
    from atm_raytracer_amd import _abi, generators, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    dist = None
    # under torch.distributed.run the process group (RCCL) is always created, so a 1-rank launch exercises the same
    # all-gather code path as N ranks; a plain `python bench.py` run has no process group
    distributed = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # ATMRT_BENCH_BACKEND=gloo is a rehearsal aid for a one-GPU box (several ranks share cuda:0 and the planes are gathered
        # through host memory); the driver's multi-GPU runs use the default, RCCL
        backend = os.environ.get("ATMRT_BENCH_BACKEND", "nccl")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the ray-marching library has no CPU path")
        if backend == "gloo":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        import datetime
        # a rank that never arrives must end the job, not hang it: torch's watchdog aborts after the time-out
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("ATMRT_BENCH_TIMEOUT", "900"))))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-marching library has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    W, H = args.width, args.height
    if W < world:
        raise SystemExit(f"bench.py: --width {W} leaves some of the {world} ranks without a pixel column")
    # pixel-column tiles, assigned INSIDE the library (csrc/atmrt_multi.hip): rank g owns [g W / G, (g + 1) W / G)
    c0, c1 = rank * W // world, (rank + 1) * W // world
    wl = c1 - c0

    t_setup = time.perf_counter()
    cfg, tiles = synth.scene(args.scene, W, H, generator=args.generator, level=args.dted_level, step=args.step)
    cfg.params.terrain_alpha = args.terrain_alpha
    if args.objects:
        synth.add_objects(cfg, n_cyl=args.objects * 7 // 10, n_bill=args.objects - args.objects * 7 // 10)
    ctx = generators.Context(local_rank)
    if distributed:
        # The frame's exchange is the library's own (C++, below the C ABI): one ncclAllGather of the tile's 84 B/pixel slab + the
        # permutation kernel into the [H][W] planes, inside atmrt_generate_image_device.  torch.distributed only carries the
        # 128-byte RCCL id to the ranks (and the timing reductions below).
        if backend == "gloo":
            # rehearsal on one GPU: RCCL refuses two ranks on one device, so the library's transport hook moves the host-staged
            # tiles over gloo; shard assignment, slab layout, assembly kernels and list gathering are the C++ code of the real run
            def all_gather(send, recv):
                dist.all_gather_into_tensor(torch.frombuffer(recv, dtype=torch.uint8), torch.frombuffer(send, dtype=torch.uint8))
            ctx.comm_init_external(rank, world, all_gather)
        elif os.environ.get("ATMRT_BENCH_TRANSPORT") == "torch":
            # the library's exchange and assembly, but the bytes moved by torch.distributed's communicator (device tensors)
            ctx.comm_init_external_device(rank, world, generators.torch_device_all_gather(dist, dev))
        else:
            # Every step of the bootstrap is agreed on by ALL ranks before any of them enters a collective of the library's own
            # communicator: (1) can RCCL be loaded here? — a local, non-collective check, MIN-reduced; (2) rank 0's id, with its
            # own ok flag; (3) ncclCommInitRank + the probe all-gather the library runs inside it (with a time-out), MIN-reduced.
            # Any no: every rank hands its device buffers to torch.distributed's communicator instead.
            ident = torch.zeros(_abi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
            ok = torch.tensor([int(ctx.lib.atmrt_comm_available())], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()):
                try:
                    if rank == 0:
                        ident.copy_(torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8))
                except Exception as exc:
                    log(f"[rank {rank}] atmrt_comm_unique_id failed ({exc})")
                    ok.zero_()
                dist.broadcast(ok, 0)
                dist.broadcast(ident, 0)
            else:
                log(f"[rank {rank}] RCCL cannot be loaded by the library on some rank")
            if int(ok.item()):
                try:
                    ctx.comm_init_rank(bytes(ident.cpu().numpy()), rank, world)
                except Exception as exc:
                    log(f"[rank {rank}] atmrt_ctx_comm_init_rank failed ({exc})")
                    ok.zero_()
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # one rank without its communicator: nobody uses it
            if not int(ok.item()):
                log(f"[rank {rank}] falling back to torch.distributed's communicator for the exchange")
                ctx.close()  # a context holds at most one communicator: start over
                ctx = generators.Context(local_rank)
                ctx.comm_init_external_device(rank, world, generators.torch_device_all_gather(dist, dev))
    terrain = generators.Terrain.from_tiles(tiles, ctx)
    log(f"[rank {rank}] scene ready in {time.perf_counter() - t_setup:.1f} s: {W}x{H}, columns [{c0},{c1}), "
        f"{len(tiles)} tiles of {next(iter(tiles.values())).shape}")

    # every rank ends the step with the WHOLE frame in its HBM: [H][W] planes (+ the lists of a multi-hit frame)
    image, pod = generators.image_planes(H, W, dev)
    local = {k: (v[..., c0:c1]) for k, v in image.items()}  # this rank's own columns of the frame

    gather_ms, comm_info = {}, {}
    multi_hit = args.terrain_alpha < 1.0 or args.objects > 0
    gathered_hits = [None]

    def make_step(generator_name):
        cfg.params.generator = _abi.GENERATORS[generator_name]
        gen = generators.make_generator(generators.Params(cfg), terrain)
        lists = multi_hit or generator_name == "InterpolatingRectilinear"

        def step():
            # returns after the library's stream has drained: tile -> slab -> ONE all-gather -> [H][W] planes
            steps, _ms = gen.generate_image_device(pod)
            tm = gen.last_timings()
            if distributed:
                ct = ctx.comm_timings()
                tm["gather_ms"] = ct["gather_ms"] + ct["assemble_ms"]
                comm_info.update(ct)
                if lists and multi_hit:  # pixels with several trace points: the variable-length lists as well (1 more collective)
                    gathered_hits[0] = gen.image_hits_device(H, W)
                    comm_info.update(ctx.comm_timings())
            tm["terrain_lookups"] = gen.last_stats()["terrain_lookups"]
            return steps, tm
        return step

    def timed(generator_name, k_steps, warmup):
        step = make_step(generator_name)
        for _ in range(warmup):
            step()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marched, phase = 0, []
        for _ in range(k_steps):
            s, tm = step()
            marched += s
            phase.append(tm)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        stats = torch.tensor([elapsed, float(marched)], dtype=torch.float64, device=dev)
        if distributed:
            tmax = stats[:1].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tot = stats[1:].clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            elapsed, marched = float(tmax.item()), float(tot.item())
            # the exchange step on its own: mean all-gather time per frame, slowest rank
            g = torch.tensor([float(np.mean([p["gather_ms"] for p in phase]))], dtype=torch.float64, device=dev)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gather_ms[generator_name] = float(g.item())
        return elapsed, marched, phase

    def counters_for(kernel):
        """rocprofv3 counter summaries committed under profiles/ (counters cannot be read from inside the process): quoted only
        when they were collected from the sources this library is built from (source_hash) and for this workload."""
        from atm_raytracer_amd import _lib
        out = {}
        if world != 1 or (W, H) != (4096, 2048) or args.scene != "headline" or args.objects or args.terrain_alpha != 1.0:
            return out
        if os.environ.get("ATMRT_LIB"):  # an experimental build: nothing cached was collected from it
            return {"sq_stale": "ATMRT_LIB is set: cached counters are never quoted for a development build"}
        built_from = _lib.build_info()["source_hash"]  # of the library that is LOADED, embedded at build time
        for key, name in (("hbm", "pmc_hbm_latest.json"), ("sq", "sq_counters_latest.json")):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            doc = json.load(open(path))
            meta = doc.get("_meta", {})
            if meta.get("source_hash") != built_from:
                out[key + "_stale"] = f"profiles/{name} was collected from sources {meta.get('source_hash')}, the loaded library was built from {built_from}"
                continue
            for k, v in doc.items():
                if kernel in k:
                    out[key] = dict(v, file=f"profiles/{name}", source_hash=meta["source_hash"], collected=meta.get("collected"))
        return out

    def roofline(generator_name, phase):
        """Dominant (longest) kernel of the generator over its mean launch duration from the library's HIP events.
        HBM side: ALGORITHMIC bytes: 8 B per terrain sample actually evaluated (the 4 int16 posts of one bilinear
        lookup; the march skips the samples of rays above every post) + what the kernel stores per pixel; the Fast
        scan is credited 8 B per ray-step, the path kernel 16 B per step of every row, the terrain profile 16 B per sample.
        Compute side (k_rect_march is FP64-VALU bound, not HBM bound): `achieved` is a real FLOP rate — FP64 flops per ray-step from
        the SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 counters of the committed rocprofv3 passes of this same command (FMA = 2 flops, add /
        mul / transcendental = 1; quoted only when collected from this build) x the ray-steps per second measured live — against
        the 78.6 TFLOP/s FP64 vector peak.  `issue_slot_frac` is the other view: ALL VALU lane-instructions per second (moves,
        compares, integer index arithmetic included) against the issue peak of 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz."""
        mean = lambda k: float(np.mean([p[k] for p in phase]))
        # this rank's tile in the last frame: the library re-cuts the tiling when the tiles' times differ (atmrt_ctx_tile_columns)
        c0, c1 = ctx.tile_columns() if distributed else (0, W)
        wl = c1 - c0
        local = {k: (v[..., c0:c1]) for k, v in image.items()}
        steps_per_launch = mean("ray_steps")
        lookups = mean("terrain_lookups")
        hits = float((local["hit_count"] > 0).sum().item())  # recorded crossings of this rank's columns in the last frame (opaque terrain: one per hit pixel)
        n_t = float(int(np.ceil(cfg.params.frame.max_distance / cfg.params.simulation_step)))  # samples per ray (utils.rs:191-196)
        n_path = n_t + 3.0                                                                      # path elements per row (utils.rs:160-170)
        per_kernel = {
            "march_ms": ("k_rect_march", 8.0 * lookups + (8 + 8 + 4 + 4) * wl * H + 32.0 * hits),
            "intersect_ms": ("k_fast_intersect", 8.0 * steps_per_launch + 8.0 * wl * H),
            "paths_ms": ("k_fast_paths", 16.0 * n_path * H),
            "profile_ms": ("k_terrain_profile", 16.0 * n_t * wl),
        }
        keys = ["march_ms"] if generator_name == "Rectilinear" else ["intersect_ms", "paths_ms", "profile_ms"]

        def kernel_entry(k):
            """ms + what bounds the kernel.  The Fast scan is credited 8 B per ray-step by SURVEY 8(d), which exceeds the HBM peak (the
            profile is re-read from L2 / Infinity Cache, not from HBM): its meaningful figure is VALU issue — instructions per
            ray-step from the cached counters of this build — so the byte credit is given as bytes, not as a bandwidth."""
            name, algo = per_kernel[k]
            e = {"ms": mean(k)}
            if k == "intersect_ms":
                sqc = counters_for(name).get("sq") or {}
                ipr_k = sqc.get("valu_lane_instructions_per_ray_step")
                e["algorithmic_bytes_credited"] = algo
                if ipr_k:
                    e["valu_lane_instructions_per_ray_step"] = ipr_k
                    e["issue_slot_frac"] = ipr_k * steps_per_launch / (mean(k) * 1e-3) / FP64_ISSUE_PEAK
            else:
                e["algorithmic_GBps"] = algo / (mean(k) * 1e-3) / 1e9
            return e
        key = max(keys, key=mean)
        kernel, algo_bytes = per_kernel[key]
        ms = mean(key)
        achieved = algo_bytes / (ms * 1e-3) / 1e9
        cached = counters_for(kernel)
        hbm_c = cached.get("hbm")
        traffic = hbm_c.get("hbm_bytes_per_frame_fetch_x2", hbm_c.get("hbm_bytes_per_launch_fetch_x2")) if hbm_c else None
        hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes,
               "traffic_source": {"cached": True, "file": hbm_c["file"], "source_hash": hbm_c["source_hash"], "collected": hbm_c.get("collected")} if hbm_c else cached.get("hbm_stale")}
        out = dict(hbm)
        sq = cached.get("sq")
        ipr = sq.get("valu_lane_instructions_per_ray_step") if sq else None
        if kernel == "k_rect_march" and ipr:
            steps_per_s = steps_per_launch / (ms * 1e-3)
            rate = ipr * steps_per_s
            flops = sq.get("fp64_flops_per_ray_step")
            valu = {"cached": True, "file": sq["file"], "source_hash": sq["source_hash"], "collected": sq.get("collected"),
                    "valu_lane_instructions_per_ray_step": ipr, "busy_frac": sq.get("valu_busy_frac"),
                    "lane_utilisation": sq.get("lane_utilisation"), "lane_instructions_per_s": rate,
                    "fp64_issue_peak_per_s": FP64_ISSUE_PEAK, "fp64_flops_per_ray_step": flops,
                    "fp64_flops_per_ray_step_hw_counter": sq.get("fp64_flops_per_ray_step_hw_counter"),
                    "effective_clock_ghz": sq.get("effective_clock_ghz"),  # GRBM_GUI_ACTIVE / 8 XCDs / the kernel's mean duration under rocprofv3
                    "lane_instructions_per_ray_step_by_kind": sq.get("lane_instructions_per_ray_step_by_kind")}
            if flops:
                kinds = sq.get("lane_instructions_per_ray_step_by_kind") or {}
                fp64_share = sum(v for k, v in kinds.items() if k.endswith("_F64")) / ipr
                out = {"bound": "fp64_valu", "achieved": flops * steps_per_s / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                       "frac": flops * steps_per_s / 1e12 / FP64_VALU_PEAK_TF, "issue_slot_frac": rate / FP64_ISSUE_PEAK, "traffic": traffic,
                       "note": "achieved = FP64 flops per ray-step (SQ_INSTS_VALU_ADD/MUL/FMA/TRANS_F64 of this build's cached rocprofv3 passes; "
                               "FMA = 2) x live ray-steps/s; issue_slot_frac = all VALU lane-instructions/s over the FP64 issue peak: the "
                               f"pipes are that full, of which {100.0 * fp64_share:.0f} % FP64 arithmetic — the rest is table-index integer work, compares, moves",
                       "valu": valu, "hbm": hbm}
            else:  # counters from before the instruction-mix passes existed: issue slots only, labelled as such
                out = {"bound": "fp64_valu_issue", "achieved": rate / 1e12, "peak": FP64_ISSUE_PEAK / 1e12, "unit": "T lane-instructions/s",
                       "frac": rate / FP64_ISSUE_PEAK, "issue_slot_frac": rate / FP64_ISSUE_PEAK, "traffic": traffic, "valu": valu, "hbm": hbm}
        elif "sq_stale" in cached:
            out["valu"] = {"stale": cached["sq_stale"]}
        if kernel == "k_rect_march" and args.objects == 0 and args.terrain_alpha == 1.0 and (wl * H + 255) // 256 <= 16384:
            # a launch of at most 16384 workgroups over opaque terrain (a column tile of a multi-GPU frame) runs the time-sliced
            # march; march_ms covers both of its kernels and the host round trip between them
            out["kernel_variant"] = ("time-sliced: k_rect_march_first + k_rect_march_cont (profiles/r03/shard_march_profile_8.json); "
                                     "the cached counters are k_rect_march's — the same loop without the slice state")
        out.update({"kernel": kernel, "kernel_ms": ms, "terrain_samples_per_launch": lookups if kernel == "k_rect_march" else None,
                    "phase_ms": {k: mean(k) for k in phase[0] if k.endswith("_ms")},
                    "all_kernels": {per_kernel[k][0]: kernel_entry(k) for k in keys if mean(k) > 0}})
        return out

    elapsed, marched, phase = timed(args.generator, args.steps, args.warmup)
    result = {
        "metric": "ray-steps/sec/GPU (and Mpixels/sec) at 4096x2048, step=100 m, max_dist=200 km",
        "value": marched / elapsed, "unit": "ray-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.scene} {W}x{H} panorama, {len(tiles)} synthetic DTED level-{args.dted_level} tiles, step {cfg.params.simulation_step:g} m, "
                               f"max_distance {cfg.params.frame.max_distance / 1000:g} km, spherical Earth + US-76 refraction, fov {cfg.params.frame.fov:g}, "
                               f"generator {args.generator}", "scene": args.scene,
                   "generator": args.generator, "width": W, "height": H, "objects": args.objects, "terrain_alpha": args.terrain_alpha, "parallelism": f"pixel-column tiles x{world}" if world > 1 else "single GPU"},
        "mpixels_per_s": W * H * args.steps / elapsed / 1e6,
        "ray_steps_per_frame": marched / args.steps,   # marched under the reference's termination rule (the `value`)
        "ray_steps_nominal_per_frame": float(W) * H * float(int(cfg.params.frame.max_distance / cfg.params.simulation_step)),  # if no ray terminated early
        "value_per_gpu": marched / elapsed / world,
        "roofline": roofline(args.generator, phase),
    }
    if distributed:
        # inside ms_per_step: one all-gather of the 84 B/pixel slab + the permutation into the [H][W] image (+ the lists of a multi-hit frame),
        # all of it in C++ below the C ABI (csrc/atmrt_multi.hip); the figures are the library's own HIP events, slowest rank
        result["all_gather_ms_per_step"] = gather_ms.get(args.generator)
        result["all_gather_collectives_per_step"] = comm_info.get("collectives")  # as the library counted them: 1 (+ 1 for the lists)
        result["tile_columns"] = list(ctx.tile_columns())  # this rank's tile in the last frame (the library re-cuts unbalanced tilings)
        result["world_size_seen"] = dist.get_world_size()
        result["comm"] = {k: comm_info.get(k) for k in ("route", "world", "bytes_per_rank", "gather_ms", "assemble_ms", "tile_ms_max")}
    if not args.only and args.generator == "Rectilinear":
        # the reference's other two generators on the same workload (secondary lines; `value` above is the per-pixel march)
        e2, m2, ph2 = timed("Fast", args.steps, 1)
        result["fast"] = {"value": m2 / e2, "unit": "ray-steps/s", "ms_per_step": e2 / args.steps * 1e3,
                          "mpixels_per_s": W * H * args.steps / e2 / 1e6, "roofline": roofline("Fast", ph2),
                          "note": "reference's default generator (params.rs:427-429): per-column terrain profile + per-row ray path"}
        e3, m3, ph3 = timed("InterpolatingRectilinear", args.steps, 1)
        result["interpolating_rectilinear"] = {"value": m3 / e3, "unit": "ray-steps/s", "ms_per_step": e3 / args.steps * 1e3,
                                               "mpixels_per_s": W * H * args.steps / e3 / 1e6,
                                               "note": "lattice of Fast-style pixels + 4-corner blend (interpolating_rectilinear.rs)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(tiles, args.dted_level)
        except Exception as exc:  # the oracle is a checker, never a fallback: report, do not hide
            result["cpu_baseline"] = {"value": None, "error": repr(exc)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed and os.environ.get("ATMRT_BENCH_CHECK_GATHER"):
        # every rank marches one more frame of the headline generator (collective), then rank 0 computes the same frame on a plain
        # single-device context and compares what the exchange left in its HBM: every plane, every trace point, every offset
        make_step(args.generator)()
        if rank == 0:
            solo = generators.Context(local_rank)
            cfg.params.generator = _abi.GENERATORS[args.generator]
            want = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, solo)).generate()
            solo.close()
            hc = torch.from_numpy(want["hit_count"].astype(np.int32)).to(dev)
            same = bool(torch.equal(image["hit_count"], hc))
            for k in ("azimuth", "elevation_angle"):
                same = same and bool(torch.equal(image[k], torch.from_numpy(want[k]).to(dev)))
            first = torch.from_numpy(want["hit_offset"].astype(np.int64)).to(dev)[hc > 0]
            for k in ("lat", "lon", "distance", "elevation", "path_length"):
                same = same and bool(torch.equal(image[k][hc > 0].view(torch.int64), torch.from_numpy(want[k]).to(dev)[first].view(torch.int64)))
            n_lists = None
            if gathered_hits[0] is not None:
                got = gathered_hits[0]
                n_lists = int(got["lat"].shape[0])
                same = same and n_lists == want["n_hits"] and bool(torch.equal(got["hit_offset"], torch.from_numpy(want["hit_offset"].astype(np.int64)).to(dev)))
                for k in ("lat", "lon", "distance", "elevation", "path_length", "normal", "rgba"):
                    same = same and bool(torch.equal(got[k].view(torch.int64), torch.from_numpy(want[k]).to(dev).view(torch.int64)))
                same = same and bool(torch.equal(got["color_tag"], torch.from_numpy(want["color_tag"].astype(np.int32)).to(dev)))
            log(f"gathered image check: {W}x{H} over {world} ranks via {comm_info.get('route')}, {int((hc > 0).sum())} hit pixels, "
                f"{n_lists if n_lists is not None else 'no'} listed trace points; matches the single-context frame: {same}")
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
