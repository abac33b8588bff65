#!/usr/bin/env python3
"""Render a panorama PNG on the GPU: generator (C ABI) -> renderer::draw_image on the device -> PNG via Pillow.
Usage: python examples/render_png.py OUT.png [--scene S3] [--width 1024] [--height 512] [--generator Fast] [--objects N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atm_raytracer_amd import config, generators, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--scene", default="S3")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--generator", default="Fast")
    ap.add_argument("--objects", type=int, default=0)
    ap.add_argument("--terrain-alpha", type=float, default=1.0)
    ap.add_argument("--coloring", default="Shading", choices=["Shading", "Simple"])
    ap.add_argument("--fog", type=float, default=None)
    a = ap.parse_args()
    from PIL import Image
    cfg, tiles = synth.scene(a.scene, a.width, a.height, generator=a.generator, step=100.0, terrain_alpha=a.terrain_alpha)
    view = {"coloring": {a.coloring: {"water_level": 300.0} if a.coloring == "Simple" else {"water_level": 300.0, "light_dir": 40.0, "light_zenith_angle": 55.0}}}
    if a.fog:
        view["fog_distance"] = a.fog
    cfg.coloring = config._coloring(view)
    if a.objects:
        synth.add_objects(cfg, n_cyl=a.objects * 7 // 10, n_bill=a.objects - a.objects * 7 // 10, dist=(2_000.0, 60_000.0),
                          radius=(40.0, 200.0), height=(200.0, 900.0), bill_w=(200.0, 800.0), bill_h=(200.0, 800.0))
    ctx = generators.Context(0)
    terrain = generators.Terrain.from_tiles(tiles, ctx)
    gen = generators.make_generator(generators.Params(cfg), terrain)
    res = gen.generate()
    rgb = generators.draw_image(ctx, generators.into_coloring(ctx.lib, cfg.params, cfg.coloring), a.width, a.height)
    Image.fromarray(rgb, "RGB").save(a.out)
    print(f"{a.out}: {a.width}x{a.height}, {res['n_hits']} trace points, {res['ray_steps']} ray-steps, {res['device_ms']:.2f} ms on device")


if __name__ == "__main__":
    main()
