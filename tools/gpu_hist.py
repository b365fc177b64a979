"""GPU box: per-ray traversal-length distribution and wave lane utilisation on the C3 workload (needs a -DPT_HIST build:
tools/build_variants.sh hist "-DPT_HIST"; PT_LIB=vk_raytrace_amd/variants/libptmi_hist.so python tools/gpu_hist.py)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vk_raytrace_amd import capi, workloads, host_device as hd
from vk_raytrace_amd.renderer import HipRenderer
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
which = sys.argv[2] if len(sys.argv) > 2 else "c3"   # c3 | c5 (PT_TUNE=accel=two for the two-level structure)
W, H = (1920, 1080) if which == "c3" else (3840, 2160)
wl = workloads.c3_sponza(W, H, 8) if which == "c3" else workloads.c5_bistro()
if os.environ.get("PT_OPAQUE_FOLIAGE") == "1":
    for m in wl.scene.materials:
        m["alphaMode"] = 0
wl.scene.finalize(capi.pack_vertices)
r = HipRenderer(); r.setup(0); r.set_scene(wl.scene); integ, _ = r.set_env(wl.env)
r.set_camera(capi.camera_lookat(wl.scene.camera, W / H)); r.set_sunsky(hd.default_sun_and_sky()); r.create((W, H))
st = hd.default_rtx_state(); st.size[0], st.size[1] = W, H; st.maxDepth = 8; st.fireflyClampThreshold = 4 * integ
L = capi.lib()
L.pt_debug_hist.restype = C.c_int
L.pt_debug_hist.argtypes = [C.c_void_p, C.c_int]
h = np.zeros((8, 40), np.uint64)
L.pt_debug_hist(h.ctypes.data, 1)
for f in range(frames):
    st.frame = f; r.setPushContants(st); r.run()
r.synchronize()
L.pt_debug_hist(h.ctypes.data, 1)
names = ["raw_all", "raw_nonopaque", "closest", "shadow", "count"]
for m in range(5):
    rays = int(h[m, 34])
    if not rays:
        continue
    it, wmax64, waves, wmax = int(h[m, 32]), int(h[m, 33]), int(h[m, 35]), int(h[m, 36])
    print(f"{names[m]:14s} rays/frame {rays / frames / 1e6:7.3f}M  iters/ray {it / rays:7.1f}  mean wave max {wmax / waves:8.1f}  lane utilisation {it / max(wmax64, 1):.3f}  wave-iterations/frame {wmax / frames / 1e6:.2f}M")
    tot = h[m, :32].sum()
    cum = 0
    row = []
    for b in range(32):
        if h[m, b]:
            row.append(f"<{1 << b}:{100.0 * int(h[m, b]) / int(tot):.2f}%")
    print("   ", " ".join(row))
for row, nm in ((5, "closest_p"), (6, "shadow_p")):
    it = int(h[row, 0])
    if it:
        print(f"{nm}: wave-iterations/frame {it / frames / 1e6:.3f}M  inner lanes/iter {int(h[row, 1]) / it:.1f}  leaf lanes/iter {int(h[row, 2]) / it:.1f}  "
              f"service rounds/frame {int(h[row, 3]) / frames / 1e3:.1f}k  iterations with inner {int(h[row, 5]) / it:.2f} leaf {int(h[row, 6]) / it:.2f} both {int(h[row, 4]) / it:.2f}")
if int(h[7, 2]):
    print(f"packet: waves/frame {int(h[7, 2]) / frames / 1e3:.1f}k  inner visits/wave {int(h[7, 0]) / int(h[7, 2]):.1f}  leaf visits/wave {int(h[7, 1]) / int(h[7, 2]):.1f}")
if int(h[7, 6]):
    print(f"shadow packet: waves/frame {int(h[7, 6]) / frames / 1e3:.1f}k  lanes in packet {int(h[7, 7]) / int(h[7, 6]):.1f}  inner visits/wave {int(h[7, 4]) / int(h[7, 6]):.1f}  leaf visits/wave {int(h[7, 5]) / int(h[7, 6]):.1f}")
if int(h[7, 12]):
    n = int(h[7, 12])
    print(f"two-level packet: waves/frame {n / frames / 1e3:.1f}k  TLAS nodes/wave {int(h[7, 8]) / n:.1f}  instances entered/wave {int(h[7, 9]) / n:.1f} (merged block {int(h[7, 14]) / n:.2f}, "
          f"sign-incoherent inside {int(h[7, 13]) / n:.2f})  BLAS nodes/wave {int(h[7, 10]) / n:.1f}  leaves/wave {int(h[7, 11]) / n:.1f}")
