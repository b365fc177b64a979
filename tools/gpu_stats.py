"""GPU box: per-ray BVH work counters (needs a -DPT_STATS build) and stage times on the C3 workload."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk_raytrace_amd import capi, workloads, host_device as hd
from vk_raytrace_amd.renderer import HipRenderer
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = workloads.c3_sponza(1920, 1080, 8, tex_size=int(os.environ.get("PT_TEX", "1024")))
if os.environ.get("PT_OPAQUE_FOLIAGE") == "1":   # experiment: how much does stochastic alpha cost?
    for m in wl.scene.materials:
        m["alphaMode"] = 0
if os.environ.get("PT_NO_TEXTURES") == "1":     # experiment: what do the texture fetches of k_shade cost? (alpha textures kept)
    for m in wl.scene.materials:
        for k in ("pbrMetallicRoughnessTexture", "emissiveTexture", "normalTexture", "transmissionTexture", "thicknessTexture", "clearcoatTexture", "clearcoatRoughnessTexture"):
            m[k] = -1
        if int(m["alphaMode"]) == 0:
            m["pbrBaseColorTexture"] = -1
wl.scene.finalize(capi.pack_vertices)
r = HipRenderer(); r.setup(0); r.set_scene(wl.scene); integ, _ = r.set_env(wl.env)
r.set_camera(capi.camera_lookat(wl.scene.camera, 1920 / 1080)); r.set_sunsky(hd.default_sun_and_sky()); r.create((1920, 1080))
st = hd.default_rtx_state(); st.size[0], st.size[1] = 1920, 1080; st.maxDepth = 8; st.fireflyClampThreshold = 4 * integ
import time
prof = os.environ.get("PT_PROF", "1") == "1"
for f in range(2):
    st.frame = f; r.setPushContants(st); r.run()
r.synchronize(); r.reset_stats()
r.set_profiling(prof)
t0 = time.perf_counter()
for f in range(2, 2 + frames):
    st.frame = f; r.setPushContants(st); r.run()
r.synchronize()
wall = (time.perf_counter() - t0) / frames * 1e3
s = r.stats()
print("WALL ms/frame %.3f  -> %.1f Msamples/s" % (wall, 1920 * 1080 / wall / 1e3))
print(json.dumps(s))
rays = s["closestRays"] + s["shadowRays"]
print("per frame: closest %.2fM shadow %.2fM alpha %.2fM" % (s["closestRays"] / frames / 1e6, s["shadowRays"] / frames / 1e6, s["alphaTests"] / frames / 1e6))
if s["nodesVisited"]:
    print("nodes per ray %.1f  tris per ray %.1f (rays = closest+shadow, re-traversals included in the numerator)" % (s["nodesVisited"] / rays, s["trisTested"] / rays))
print("ms/frame: closest %.2f shade %.2f shadow %.2f gen %.2f acc %.2f" % tuple(s[k] / frames for k in ("msTraceClosest", "msShade", "msTraceShadow", "msGenerate", "msAccumulate")))
