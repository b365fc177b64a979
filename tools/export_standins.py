"""Writes the synthetic stand-in scenes as glTF 2.0 (.glb) so that the reference application itself can render them wherever
Vulkan exists (the cross-check BASELINE.md section 4 asks for): python tools/export_standins.py <out_dir> [c1 c2 c3 c5 feature]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk_raytrace_amd import gltf, synth, workloads
out = sys.argv[1]
which = sys.argv[2:] or ["c1", "feature", "c2", "c3"]
os.makedirs(out, exist_ok=True)
makers = {"c1": lambda: workloads.c1_quad().scene, "feature": lambda: synth.feature_box(), "c2": lambda: workloads.c2_helmet().scene,
          "c3": lambda: workloads.c3_sponza().scene, "c5": lambda: workloads.c5_bistro().scene}
for w in which:
    sc = makers[w]()
    p = gltf.save_gltf(sc, os.path.join(out, f"standin_{w}.glb"))
    print(w, p, os.path.getsize(p) // 1024, "KiB", sc.num_triangles, "triangles")
