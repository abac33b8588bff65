"""Command line mirroring the reference's subcommands (src/main.rs:17-39) over the C ABI:

    python -m atm_raytracer_amd gen -c CONFIG.yaml [--output OUT.png] [--metadata OUT.npz|OUT.dat]
    python -m atm_raytracer_amd output-atm CONFIG.yaml [-a MIN] [-b MAX] [-s STEP] [-c]
    python -m atm_raytracer_amd output-ray-paths CONFIG.yaml [-h H] [-a MIN] [-b MAX] [-s DEG] [-r STEP] [-c CUTOFF] [-o OUTSTEP]
    python -m atm_raytracer_amd output-elev-profile CONFIG.yaml [-a AZIM] [-s STEP] [-c CUTOFF]

Column formats follow src/atm_printer.rs:37-46, src/ray_path.rs:65-103 and src/elev_profile.rs:43-64.  `gen` writes the
image of renderer::draw_image (no ticks / labels: those stay CPU-side in the reference, renderer/mod.rs:28-323) and, on
request, the per-pixel metadata as a compressed .npz (the reference's bincode+gzip layout depends on crates that are absent).
Floats are printed with Python's repr, the shortest round-trip form like Rust's `{}`.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

from . import _abi, config, generators


def _ctx_with_terrain(cfg, cfg_path):
    ctx = generators.Context()
    folder = os.path.join(os.getcwd(), cfg.terrain_folder)  # env::current_dir().push(terrain_folder), generator/mod.rs:57-58
    terrain = generators.Terrain.from_folder(folder, ctx)
    return ctx, terrain


def _configure(ctx, cfg):
    ctx.check(ctx.lib.atmrt_set_params(ctx.handle, C.byref(cfg.params)))
    ctx.check(ctx.lib.atmrt_set_atmosphere(ctx.handle, C.byref(cfg.atmosphere)))


def cmd_gen(a):
    start = time.time()
    cfg = config.parse_config(a.config)
    stamp = lambda msg: print(f"{time.time() - start:.3f}: {msg}", flush=True)
    stamp(f"Using terrain data directory: {os.path.join(os.getcwd(), cfg.terrain_folder)!r}")
    ctx, terrain = _ctx_with_terrain(cfg, a.config)
    print(f"Detected {terrain.n_files} terrain files")
    gen = generators.make_generator(generators.Params(cfg), terrain)
    stamp("Calculating pixels...")
    res = gen.generate()
    stamp("Done calculating")
    stamp("Outputting image...")
    col = generators.into_coloring(ctx.lib, cfg.params, cfg.coloring)
    rgb = generators.draw_image(ctx, col, res["width"], res["height"])
    from PIL import Image
    Image.fromarray(rgb, "RGB").save(a.output)
    meta_path = a.metadata or cfg.output["file_metadata"]  # `if let Some(ref filename) = params.output.file_metadata`, generator/mod.rs:88-94
    if meta_path:
        stamp("Outputting metadata...")
        if meta_path.endswith(".npz"):  # this package's own array dump
            np.savez_compressed(meta_path, **{k: v for k, v in res.items() if isinstance(v, np.ndarray)})
        else:  # gzip(bincode(AllData)) in the reference's field order — but with a stand-in `env` segment (metadata.py): this package's
            # reader only; write_metadata warns on stderr every time
            from . import metadata
            metadata.write_metadata(meta_path, cfg, res, col, metadata.object_elevations(cfg, terrain))
    stamp("Done.")
    return 0


def cmd_output_atm(a):
    cfg = config.parse_config(a.config)
    ctx = generators.Context()
    _configure(ctx, cfg)
    assert a.step > 0
    alts, alt = [], a.min_alt
    while alt <= a.max_alt:  # atm_printer.rs:37-46
        alts.append(alt)
        alt += a.step
    s = generators.atmosphere_sample(ctx, alts)
    for h, t, p in zip(alts, s["temperature"], s["pressure"]):
        print(f"{h!r} {float(t - (273.15 if a.celsius else 0.0))!r} {float(p)!r} 0.0")  # dry air: humidity 0
    return 0


def cmd_output_ray_paths(a):
    cfg = config.parse_config(a.config)
    ctx = generators.Context()
    _configure(ctx, cfg)
    assert a.angle_step > 0, "step must be positive"  # ray_path.rs:53
    angs, ang = [], a.min_ang
    while ang <= a.max_ang:
        angs.append(ang)
        ang += a.angle_step
    n_steps = int(np.ceil(a.cutoff / a.ray_step)) + 1
    x, h = generators.ray_paths(ctx, a.height, angs, a.ray_step, n_steps, straight=False)
    # ray_path.rs:76-91: keep a sample whenever the step straddles a multiple of output_step; stop after x >= cutoff
    xs, keep = [0.0], [0]
    for k in range(1, n_steps + 1):
        xv = x[0, k]
        if np.floor((xv - a.ray_step / 2.0) / a.output_step) != np.floor((xv + a.ray_step / 2.0) / a.output_step):
            xs.append(float(xv))
            keep.append(k)
        if xv >= a.cutoff:
            break
    for xv, k in zip(xs, keep):
        print("\t".join([repr(xv)] + [repr(float(h[i, k])) for i in range(len(angs))]) + "\t")
    return 0


def cmd_output_elev_profile(a):
    cfg = config.parse_config(a.config)
    ctx, terrain = _ctx_with_terrain(cfg, a.config)
    _configure(ctx, cfg)
    assert a.step > 0, "step must be positive"  # elev_profile.rs:32
    xs, x = [], 0.0
    while x <= a.cutoff:  # elev_profile.rs:54-60
        xs.append(x)
        x += a.step
    lat, lon = generators.coords_at_dist(ctx, cfg.params.position.latitude, cfg.params.position.longitude, a.azim, xs)
    elev, valid = terrain.get_elev(lat, lon)
    for xv, e, ok in zip(xs, elev, valid):
        print(f"{xv!r}\t{float(e) if ok else 0.0!r}")
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="atm_raytracer_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    g = sub.add_parser("gen")
    g.add_argument("-c", "--config", required=True)
    g.add_argument("--output", default="./output.png")
    g.add_argument("--metadata", default=None)
    g.set_defaults(fn=cmd_gen)
    p = sub.add_parser("output-atm")
    p.add_argument("config")
    p.add_argument("-a", "--min-alt", type=float, default=0.0)
    p.add_argument("-b", "--max-alt", type=float, default=1000.0)
    p.add_argument("-s", "--step", type=float, default=0.2)
    p.add_argument("-c", "--celsius", action="store_true")
    p.set_defaults(fn=cmd_output_atm)
    r = sub.add_parser("output-ray-paths", add_help=False)
    r.add_argument("config")
    r.add_argument("-h", "--height", type=float, default=2.0)
    r.add_argument("-a", "--min-ang", type=float, default=-1.0)
    r.add_argument("-b", "--max-ang", type=float, default=1.0)
    r.add_argument("-s", "--angle-step", type=float, default=0.1)
    r.add_argument("-r", "--ray-step", type=float, default=50.0)
    r.add_argument("-c", "--cutoff", "--cutoff-dist", type=float, default=10000.0)
    r.add_argument("-o", "--output-step", type=float, default=50.0)
    r.set_defaults(fn=cmd_output_ray_paths)
    e = sub.add_parser("output-elev-profile")
    e.add_argument("config")
    e.add_argument("-a", "--azim", type=float, default=0.0)
    e.add_argument("-s", "--step", type=float, default=50.0)
    e.add_argument("-c", "--cutoff", "--cutoff-dist", type=float, default=10000.0)
    e.set_defaults(fn=cmd_output_elev_profile)
    a = ap.parse_args(argv)
    try:
        return a.fn(a)
    except (generators.AtmrtError, config.ConfigError, OSError) as exc:
        print(f"ERROR: {exc}", file=sys.stderr)  # main.rs:36-38
        return 1


if __name__ == "__main__":
    sys.exit(main())
