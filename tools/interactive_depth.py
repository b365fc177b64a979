"""GPU box: the display loop (render one frame, pt_tonemap_begin, collect the image issued `pending` calls earlier) at several pipeline depths.
   usage: [PT_LIB=...] python tools/interactive_depth.py "inflight:pending[:displaySlots[:bands]]" ...      e.g. 4:3:0 4:5:2 6:5:0 4:0:2:6
   pending 0 = the host waits for every image (pt_tonemap); bands = PT_TUNE bands (a single frame on an idle GPU cut into bands of tiles)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk_raytrace_amd import capi, workloads, host_device as hd
from vk_raytrace_amd.renderer import HipRenderer

wl = workloads.c3_sponza(1920, 1080, 256)
wl.scene.finalize(capi.pack_vertices)
W, H = wl.width, wl.height
for rnd in range(2):
    for combo in sys.argv[1:]:
        f = [int(x) for x in combo.split(":")]
        infl, pend, disp, bands = f[0], f[1], (f[2] if len(f) > 2 else 0), (f[3] if len(f) > 3 else 1)
        os.environ["PT_TUNE"] = f"inflight={infl},displaySlots={disp},bands={bands}" + ("," + os.environ["PT_TUNE_EXTRA"] if os.environ.get("PT_TUNE_EXTRA") else "")
        r = HipRenderer(); r.setup(0); r.set_scene(wl.scene); integ, _ = r.set_env(wl.env)
        r.set_camera(capi.camera_lookat(wl.scene.camera, W / H)); r.set_sunsky(hd.default_sun_and_sky()); r.create((W, H))
        st = hd.default_rtx_state(); st.size[0], st.size[1] = W, H; st.maxDepth = 8; st.fireflyClampThreshold = 4 * integ
        tm = hd.default_tonemapper()
        frame = 0
        def loop(n):
            global frame
            for _ in range(n):
                st.frame = frame; r.setPushContants(st); r.run(); frame += 1
                if pend == 0:
                    r.tonemap(tm)
                    continue
                r.tonemap_begin(tm)
                if r.tonemap_pending() > pend:
                    r.tonemap_end()
            while r.tonemap_pending():
                r.tonemap_end()
        loop(16)
        r.synchronize()
        t0 = time.perf_counter(); loop(96); t = time.perf_counter() - t0
        print(f"inflight {infl} displaySlots {disp} bands {bands} pending {pend}: {t / 96 * 1e3:.3f} ms / displayed frame = {W * H * 96 / t / 1e6:.0f} Msamples/s", flush=True)
        r.destroy()
