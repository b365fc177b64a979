"""The benchmark configurations of BASELINE.json made concrete.

Real assets are absent on every box (the reference downloads them at configure time), so each
configuration is rendered on its seeded synthetic stand-in from ``synth``; the configuration name says so.
"""
from dataclasses import dataclass

import numpy as np

from . import synth


@dataclass
class Workload:
    name: str
    scene: object
    env: np.ndarray
    width: int
    height: int
    spp: int          # frames x maxSamples with maxSamples = 1 (reference default, src/sample_example.hpp:165)
    depth: int
    pbr_mode: int     # 0 Disney, 1 glTF
    note: str = ""


def real_asset(name):
    """A real glTF asset when the PT_ASSET_DIR environment variable points at a directory that holds it (SURVEY.md 8(d):
    "real assets used instead if PT_ASSET_DIR provides them"); None otherwise.  `name`: e.g. "Sponza" -- looked up as
    <dir>/<name>.gltf|.glb, <dir>/<name>/<name>.gltf and <dir>/<name>/glTF/<name>.gltf (the Khronos sample-model layout)."""
    import os
    root = os.environ.get("PT_ASSET_DIR")
    if not root:
        return None
    for rel in (f"{name}.gltf", f"{name}.glb", f"{name}/{name}.gltf", f"{name}/glTF/{name}.gltf", f"{name}/glTF-Binary/{name}.glb"):
        p = os.path.join(root, rel)
        if os.path.exists(p):
            # libptmi's own importer (pt_gltf_load, csrc/pt_gltf.cpp): the path a C++ host takes; PT_ASSET_IMPORTER=python selects the
            # Python importer (vk_raytrace_amd/gltf.py, its cross-check)
            if os.environ.get("PT_ASSET_IMPORTER") == "python":
                from . import gltf
                return gltf.load_gltf(p)
            from .scene import GltfFileScene
            return GltfFileScene(p)
    return None


def c1_quad():
    return Workload("C1 quad 256x256 1spp", synth.quad_scene(), synth.constant_env(16, 8, 1.0), 256, 256, 1, 10, 0)


def c2_helmet(scale=1.0):
    ts = max(64, int(2048 * scale))
    return Workload("C2 helmet-like (synthetic DamagedHelmet stand-in) 1024x1024 64spp depth4 glTF-PBR",
                    synth.helmet_like(target_tris=int(70_000 * max(scale, 0.05)), tex_size=ts),
                    synth.procedural_sky(max(64, int(2048 * scale)), max(32, int(1024 * scale))),
                    max(64, int(1024 * scale)), max(64, int(1024 * scale)), 64, 4, 1)


def c3_sponza(width=1920, height=1080, spp=256, tex_size=1024, target_tris=262_267, env_w=2048):
    real = real_asset("Sponza")
    if real is not None:
        return Workload(f"C3 Sponza (glTF file {getattr(real, 'path', 'from PT_ASSET_DIR')}, {real.num_triangles} tris) {width}x{height} {spp}spp depth8 Disney + HDR env",
                        real, synth.procedural_sky(env_w, env_w // 2), width, height, spp, 8, 0, note="gltf")
    return Workload(f"C3 sponza-like (synthetic Crytek-Sponza stand-in, {target_tris} tris target) {width}x{height} {spp}spp depth8 Disney + HDR env",
                    synth.sponza_like(target_tris=target_tris, tex_size=tex_size), synth.procedural_sky(env_w, env_w // 2), width, height, spp, 8, 0)


def c4_sponza_4k():
    w = c3_sponza(3840, 2160, 1024)
    w.name = w.name.replace("C3", "C4")
    return w


def c5_bistro(target_tris=3_800_000, tex_size=512):
    return Workload("C5 bistro-like (synthetic Bistro stand-in) 3840x2160 4096spp depth8 Disney + HDR env",
                    synth.bistro_like(target_tris=target_tris, tex_size=tex_size), synth.procedural_sky(2048, 1024), 3840, 2160, 4096, 8, 0)
