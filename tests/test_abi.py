"""The C-ABI library loads and exports every symbol include/pt_api.h declares; struct layouts match the
reference's scalar block layout; without a GPU the product fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vk_raytrace_amd import capi, host_device as hd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    L = capi.lib()
    names = declared_symbols()
    assert len(names) >= 24
    bound = {n for n, _, _ in capi.API}
    for n in names:
        assert hasattr(L, n), f"{n} declared in pt_api.h but not exported by libptmi.so"
        assert n in bound, f"{n} declared in pt_api.h but not bound in capi.API"
    assert bound <= set(names)


def test_struct_sizes_match_reference_layout():
    # SURVEY.md Appendix A (scalar block layout)
    want = {hd.RtxState: 48, hd.SceneCamera: 140, hd.VertexAttributes: 32, hd.GltfShadeMaterial: 216, hd.Light: 64, hd.EnvAccel: 16,
            hd.Tonemapper: 48, hd.SunAndSky: 96}
    for t, n in want.items():
        assert C.sizeof(t) == n, t
    m = hd.GltfShadeMaterial
    assert m.uvTransform.offset == 18 * 4 and m.ior.offset == 37 * 4 and m.sheen.offset == 52 * 4 and m.clearcoatFactor.offset == 48 * 4
    assert hd.SunAndSky.sun_direction.offset == 16 * 4 and hd.RtxState.size.offset == 8 * 4


def test_name_and_defaults():
    assert capi.lib().pt_renderer_name() == b"HIP"
    st = hd.default_rtx_state()
    assert (st.maxDepth, st.maxSamples, st.pbrMode, st.hdrMultiplier, st.maxHeatmap) == (10, 1, 0, 1.0, 65000)  # src/sample_example.hpp:162-174
    tm = hd.default_tonemapper()
    assert (tm.dither, tm.autoExposure, tm.Ywhite, tm.key) == (1, 0, 0.5, 0.5)  # src/render_output.hpp:37-49


def _has_gpu():
    ctx = C.c_void_p()
    rc = capi.lib().pt_create(0, C.byref(ctx))
    if rc == capi.PT_OK:
        capi.lib().pt_destroy(ctx)
        return True
    return False


def test_no_cpu_fallback():
    """On a box without a gfx950 device the product refuses to create a context (and says why)."""
    if _has_gpu():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = capi.lib().pt_create(0, C.byref(ctx))
    assert rc == capi.PT_ERR_NO_DEVICE and not ctx.value
    assert b"no CPU fallback" in capi.lib().pt_last_error(None) or b"gfx950" in capi.lib().pt_last_error(None)
    from vk_raytrace_amd.renderer import HipRenderer
    with pytest.raises(capi.PtError):
        HipRenderer().setup(0)


def test_null_arguments_are_errors_not_crashes():
    L = capi.lib()
    assert L.pt_destroy(None) == capi.PT_ERR_INVALID
    assert L.pt_build_accel(None) == capi.PT_ERR_INVALID
    assert L.pt_pack_vertices(4, None, None, None, None, None, None) == capi.PT_ERR_INVALID
    assert L.pt_build_env_accel(None, 4, 4, None, None, None) == capi.PT_ERR_INVALID
    cam = hd.SceneCamera()
    e = np.zeros(3, np.float32)
    assert L.pt_camera_lookat(e.ctypes.data, e.ctypes.data, e.ctypes.data, 45.0, 1.0, C.byref(cam)) == capi.PT_ERR_INVALID  # eye == center


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under vk_raytrace_amd/ may reference it."""
    pkg = os.path.join(ROOT, "vk_raytrace_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liborc" not in txt and "orc_" not in txt and "from tests" not in txt and "import tests" not in txt, os.path.join(dp, f)
                for line in txt.splitlines():
                    if "oracle" in line and ("#include" in line or "import" in line):
                        raise AssertionError(f"{f}: {line}")


def _parse_tuning(s):
    L = capi.lib()
    out = (C.c_int * 32)()
    unk = C.create_string_buffer(512)
    L.pt_debug_parse_tuning.restype = C.c_int
    n = L.pt_debug_parse_tuning(s.encode() if s is not None else None, out, 32, unk, 512)
    keys = ["stateMB", "stateGB", "packetClosest", "mergeSingles", "cnodes", "shadeTris", "tail", "warm", "texTile", "texGroups", "regen", "packetTwo", "blasWorkers", "batch",
            "inflight", "displaySlots", "bands", "bandTiles", "fuse", "build", "accel"]
    assert n == len(keys)
    return dict(zip(keys, list(out)[:n])), unk.value.decode()


def test_pt_tune_is_parsed_key_by_key():
    """PT_TUNE (read by pt_create into the CONTEXT's knobs since round 6): exact key matches, every knob reachable, unknown tokens reported -- the removed knobs
    (waves, refill, chunk, packetWaves, ...) among them, and "waves=" no longer matches inside "packetWaves="."""
    d, unk = _parse_tuning(None)
    assert unk == "" and (d["tail"], d["batch"], d["inflight"], d["build"], d["accel"], d["fuse"], d["bands"], d["bandTiles"]) == (65536, 64, 4, 3, 0, 1, 3, 64)
    d, unk = _parse_tuning("tail=0, batch=4,build=sah,accel=two,inflight=3,stateMB=12,packetClosest=2,fuse=2")
    assert unk == "" and (d["tail"], d["batch"], d["build"], d["accel"], d["inflight"], d["stateMB"], d["packetClosest"], d["fuse"]) == (0, 4, 1, 1, 3, 12, 2, 2)
    d, unk = _parse_tuning("packetWaves=7,waves=16,refill=8,chunk=64,bogus,fuse=x,build=quick,tail=5")
    assert unk == "packetWaves=7,waves=16,refill=8,chunk=64,bogus,fuse=x,build=quick" and d["tail"] == 5 and d["fuse"] == 1 and d["build"] == 3
    for k, v in (("build=lbvh", 0), ("build=sah", 1), ("build=ploc", 2), ("build=sahdev", 3)):
        assert _parse_tuning(k)[0]["build"] == v
    assert _parse_tuning("bandTiles=0")[0]["bandTiles"] == 1
    every = "stateMB=1,stateGB=2,packetClosest=3,mergeSingles=0,cnodes=0,shadeTris=0,tail=7,warm=0,texTile=0,texGroups=0,regen=0,packetTwo=0,blasWorkers=2,batch=9,inflight=2,displaySlots=1,bands=5,bandTiles=6,fuse=0"
    d, unk = _parse_tuning(every)
    assert unk == "" and [d[t.split("=")[0]] for t in every.split(",")] == [int(t.split("=")[1]) for t in every.split(",")]
