"""GPU box: flat vs two-level acceleration structure on a workload, pixel by pixel; differing pixels are re-rendered by the CPU oracle.
   python tools/gpu_two_level_diff.py c5 [frames] [depth]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import orc
from tests.common import Config
from vk_raytrace_amd import capi, workloads, host_device as hd
from vk_raytrace_amd.renderer import HipRenderer

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
wl = {"c5": workloads.c5_bistro, "c3": workloads.c3_sponza, "c2": workloads.c2_helmet}[name]()
wl.scene.finalize(capi.pack_vertices)
W, H = wl.width, wl.height
cfg = Config(wl.scene, wl.env, W, H, depth=depth, pbr=wl.pbr_mode)
r = HipRenderer(); r.setup(0); r.set_scene(cfg.scene); integral, _ = r.set_env(cfg.env); r.set_camera(cfg.camera); r.set_sunsky(cfg.sunsky); r.create((W, H))


def render(debug=0, n=frames):
    st = cfg.state(integral)
    st.debugging_mode = debug
    for f in range(n):
        st.frame = f
        r.setPushContants(st); r.run()
    return r.read_accum()


imgs = {}
for mode in (capi.PT_ACCEL_FLAT, capi.PT_ACCEL_TWO_LEVEL):
    r.set_accel_mode(mode)
    imgs[mode] = {"path": render(), "normal": render(hd.eNormal, 1), "uv": render(hd.eTexcoord, 1), "stats": r.stats()}
    r.reset_stats()
for k in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests"):
    print(k, imgs[0]["stats"][k], imgs[1]["stats"][k])
for what in ("normal", "uv", "path"):
    a, b = imgs[0][what].reshape(-1, 4), imgs[1][what].reshape(-1, 4)
    bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(1))[0]
    print(what, "differing pixels:", len(bad), [(int(i % W), int(i // W)) for i in bad[:12]])
    if len(bad) and what == "path":
        ids = bad[:64].astype(np.uint32)
        o = orc.Oracle(); o.set_scene(cfg.scene); integ, _ = o.set_env(cfg.env); o.set_camera(cfg.camera); o.set_sunsky(cfg.sunsky)
        st = cfg.state(integ)
        acc = np.zeros((H, W, 4), np.float32)
        for f in range(frames):
            st.frame = f
            o.render_frame(st, acc, ids)
        ref = acc.reshape(-1, 4)[ids]
        print("oracle == flat:", int((ref.view(np.uint32) == a[ids].view(np.uint32)).all(1).sum()), "oracle == two-level:", int((ref.view(np.uint32) == b[ids].view(np.uint32)).all(1).sum()), "of", len(ids))
        for i in range(min(6, len(ids))):
            print(int(ids[i] % W), int(ids[i] // W), "flat", a[ids[i]][:3], "two", b[ids[i]][:3], "oracle", ref[i][:3])
    if len(bad) and what != "path":
        for i in bad[:6]:
            print(int(i % W), int(i // W), "flat", a[i][:3], "two", b[i][:3])
