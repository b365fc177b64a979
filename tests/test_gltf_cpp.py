"""The C++ glTF importer (pt_gltf_load, vk_raytrace_amd/csrc/pt_gltf.cpp) against the Python one (vk_raytrace_amd/gltf.py): both must
produce the same flat scene arrays -- bit for bit on files that carry all attributes (exporter round trips), to float tolerance where
normals / tangents are synthesised or node transforms are composed -- plus PNG / JPEG decoder checks against Pillow."""
import ctypes as C
import io
import json
import os

import numpy as np
import pytest

from vk_raytrace_amd import capi, gltf, host_device as hd, synth
from tests.test_gltf import _tri_doc, _write


class CppScene:
    def __init__(self, path):
        L = capi.lib()
        self.L = L
        self.h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.pt_gltf_load(path.encode(), C.byref(self.h), err, 512)
        if rc != capi.PT_OK:
            raise ValueError(err.value.decode())
        d = L.pt_gltf_desc(self.h).contents
        self.vertices = np.ctypeslib.as_array(C.cast(d.vertices, C.POINTER(C.c_uint8)), (d.numVertices * 32,)).copy() if d.numVertices else np.zeros(0, np.uint8)
        self.indices = np.ctypeslib.as_array(C.cast(d.indices, C.POINTER(C.c_uint32)), (d.numIndices,)).copy() if d.numIndices else np.zeros(0, np.uint32)
        self.prim_meshes = np.frombuffer(C.string_at(d.primMeshes, d.numPrimMeshes * hd.primmesh_dtype.itemsize), hd.primmesh_dtype).copy()
        self.nodes = np.frombuffer(C.string_at(d.nodes, d.numNodes * hd.node_dtype.itemsize), hd.node_dtype).copy()
        self.materials = np.frombuffer(C.string_at(d.materials, d.numMaterials * hd.material_dtype.itemsize), hd.material_dtype).copy()
        self.lights = np.frombuffer(C.string_at(d.lights, d.numLights * hd.light_dtype.itemsize), hd.light_dtype).copy() if d.numLights else np.zeros(0, hd.light_dtype)
        self.textures = []
        tex = C.cast(d.textures, C.POINTER(hd.TextureDesc))
        for i in range(d.numTextures):
            t = tex[i]
            img = np.ctypeslib.as_array(C.cast(t.rgba8, C.POINTER(C.c_uint8)), (t.height, t.width, 4)).copy()
            self.textures.append((img, t.magFilter, t.minFilter, t.wrapS, t.wrapT))
        e, c, u = (np.zeros(3, np.float32) for _ in range(3))
        f = C.c_float()
        L.pt_gltf_camera(self.h, e.ctypes.data, c.ctypes.data, u.ctypes.data, C.byref(f))
        self.camera = (e, c, u, f.value)

    def close(self):
        self.L.pt_gltf_free(self.h)


def compare(py, cpp, exact=True):
    py.finalize(capi.pack_vertices)
    pv = py.vertices.view(np.uint8).reshape(-1)
    if exact:
        assert np.array_equal(pv, cpp.vertices)
    else:
        a = np.frombuffer(pv.tobytes(), hd.vertex_dtype); b = np.frombuffer(cpp.vertices.tobytes(), hd.vertex_dtype)
        assert np.allclose(a["position"], b["position"]) and np.allclose(a["texcoord"], b["texcoord"], atol=1e-6)
    assert np.array_equal(py.indices, cpp.indices)
    assert [tuple(int(x) for x in t) for t in py.prim_meshes] == [tuple(int(x) for x in t) for t in cpp.prim_meshes.tolist()]
    assert len(py.nodes) == len(cpp.nodes)
    for (m, pm), nd in zip(py.nodes, cpp.nodes):
        assert int(nd["primMesh"]) == pm
        got = np.asarray(nd["worldMatrix"]).reshape(4, 4).T
        assert np.array_equal(got, m) if exact else np.allclose(got, m, rtol=1e-6, atol=1e-6)
    assert len(py.materials) == len(cpp.materials)
    for x, y in zip(py.materials, cpp.materials):
        assert x.tobytes() == y.tobytes()
    assert len(py.textures) == len(cpp.textures)
    for x, (img, mag, mn, ws, wt) in zip(py.textures, cpp.textures):
        assert np.array_equal(x.rgba8, img) and (x.magFilter, x.minFilter, x.wrapS, x.wrapT) == (mag, mn, ws, wt)
    assert len(py.lights) == len(cpp.lights)
    for x, y in zip(py.lights, cpp.lights):
        assert x.tobytes() == y.tobytes()
    e, c, u, f = cpp.camera
    assert np.allclose(py.camera.eye, e, atol=1e-6) and np.allclose(py.camera.center, c, atol=1e-5) and np.allclose(py.camera.up, u, atol=1e-6) and abs(py.camera.fov - f) < 1e-4


@pytest.mark.parametrize("ext", ["gltf", "glb"])
@pytest.mark.parametrize("make", [lambda: synth.feature_box(tex_size=16, lights=True), lambda: synth.quad_scene(), lambda: synth.sponza_like(target_tris=3000, tex_size=8)])
def test_cpp_importer_matches_python_on_round_trips(tmp_path, make, ext):
    path = str(tmp_path / f"scene.{ext}")
    gltf.save_gltf(make(), path)
    cpp = CppScene(path)
    compare(gltf.load_gltf(path), cpp, exact=True)
    cpp.close()


def test_cpp_importer_synthesis_and_trs(tmp_path):
    doc = _tri_doc()
    doc["nodes"] = [{"children": [1, 2], "translation": [1, 2, 3]}, {"mesh": 0, "rotation": [0, 0, 0.70710678, 0.70710678], "scale": [2, 2, 2]}, {"mesh": 0}]
    doc["scenes"] = [{"nodes": [0]}]
    path = _write(tmp_path, doc)
    cpp = CppScene(path)
    py = gltf.load_gltf(path)
    compare(py, cpp, exact=False)
    a = np.frombuffer(cpp.vertices.tobytes(), hd.vertex_dtype)
    py.finalize(capi.pack_vertices)
    assert np.array_equal(a["normal"], py.vertices["normal"]) and np.array_equal(a["tangent"], py.vertices["tangent"])   # synthesised attributes, packed identically
    cpp.close()


def test_scene_selection_and_nodes_outside_the_scene(tmp_path):
    """Import rules the reference leaves to nvh::GltfScene (src/scene.cpp:56-76), pinned here on both importers: only the nodes reachable from the
    roots of the file's default scene (`scene`, 0 when absent) are instantiated -- a second scene and orphan nodes contribute nothing; a file
    without `scenes` instantiates every node that is nobody's child; a mesh used by two nodes of the scene is ONE prim-mesh with two
    instances, a mesh that no instantiated node uses is not imported at all."""
    doc = _tri_doc()
    doc["meshes"].append({"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]})      # same accessors: de-duplicated with mesh 0
    doc["nodes"] = [{"mesh": 0, "translation": [1, 0, 0]}, {"mesh": 1, "translation": [0, 2, 0]}, {"mesh": 0, "translation": [0, 0, 3]}, {"mesh": 0, "translation": [9, 9, 9]}]
    doc["scenes"] = [{"nodes": [3]}, {"nodes": [0, 1]}]
    doc["scene"] = 1
    path = _write(tmp_path, doc)
    py, cpp = gltf.load_gltf(path), CppScene(path)
    compare(py, cpp, exact=False)
    assert len(py.nodes) == 2 and len(py.prim_meshes) == 1                                     # nodes 0 and 1; node 2 (orphan) and scene 0 are ignored
    assert sorted(float(m[0][3] + m[1][3] + m[2][3]) for m, _ in py.nodes) == [1.0, 2.0]
    cpp.close()
    del doc["scene"]                                                                            # default scene = 0
    path = _write(tmp_path, doc, "t0.gltf")
    py, cpp = gltf.load_gltf(path), CppScene(path)
    compare(py, cpp, exact=False)
    assert len(py.nodes) == 1 and float(py.nodes[0][0][0][3]) == 9.0
    cpp.close()
    del doc["scenes"]                                                                           # no scenes: every root node
    path = _write(tmp_path, doc, "t1.gltf")
    py, cpp = gltf.load_gltf(path), CppScene(path)
    compare(py, cpp, exact=False)
    assert len(py.nodes) == 4 and len(py.prim_meshes) == 1
    cpp.close()


def test_cpp_importer_errors(tmp_path):
    with pytest.raises(ValueError):
        CppScene(str(tmp_path / "missing.gltf"))
    (tmp_path / "bad.gltf").write_text("{ not json")
    with pytest.raises(ValueError):
        CppScene(str(tmp_path / "bad.gltf"))
    doc = _tri_doc()
    doc["accessors"][0]["sparse"] = {"count": 1}
    with pytest.raises(ValueError, match="sparse"):
        CppScene(_write(tmp_path, doc))


def test_cpp_importer_rejects_hostile_files(tmp_path):
    """Untrusted input: every malformed offset / size / hierarchy must come back as PT_ERR_INVALID with a message, never as an
    out-of-bounds read, a signal or unbounded recursion (ASan / UBSan build: tools/asan_gltf_main.cpp)."""
    import struct
    import zlib

    def bad(mutate, match=None):
        doc = _tri_doc()
        mutate(doc)
        with pytest.raises(ValueError, match=match):
            CppScene(_write(tmp_path, doc))

    # sizes that wrap size_t when added, are negative, fractional or not finite
    bad(lambda d: d["bufferViews"][0].update(byteOffset=1e19, byteLength=48), "valid size|exceeds")
    bad(lambda d: d["bufferViews"][0].update(byteOffset=2 ** 63, byteLength=2 ** 63), "valid size|exceeds")
    bad(lambda d: d["bufferViews"][0].update(byteStride=2 ** 63), "valid size|byteStride")
    bad(lambda d: d["bufferViews"][0].update(byteStride=8), "byteStride smaller")
    bad(lambda d: d["bufferViews"][0].update(byteOffset=-4), "valid size")
    bad(lambda d: d["bufferViews"][0].update(byteLength=47.5), "valid size")
    bad(lambda d: d["accessors"][0].update(byteOffset=2 ** 62), "exceeds|valid size")
    bad(lambda d: d["accessors"][0].update(byteOffset=2 ** 40), "exceeds")
    bad(lambda d: d["accessors"][0].update(count=2 ** 40), "count too large|exceeds")
    bad(lambda d: d["accessors"][0].update(count=5), "exceeds")
    bad(lambda d: d["accessors"][0].update(sparse={"count": 2 ** 60, "indices": {"bufferView": 1, "componentType": 5123}, "values": {"bufferView": 0}}), "sparse|valid size")
    # node hierarchy with a cycle, directly and through a chain
    bad(lambda d: d.update(nodes=[{"children": [1]}, {"children": [0], "mesh": 0}]), "cycle")
    bad(lambda d: d.update(nodes=[{"children": [0], "mesh": 0}]), "cycle")
    # a diamond (one node under two parents) is legal glTF-wise for this importer: instanced twice, no error
    doc = _tri_doc()
    doc["nodes"] = [{"children": [1, 2]}, {"children": [3]}, {"children": [3]}, {"mesh": 0}]
    assert len(CppScene(_write(tmp_path, doc)).nodes) == 2
    # JSON nested beyond any sane depth
    (tmp_path / "deep.gltf").write_text("[" * 100000 + "]" * 100000)
    with pytest.raises(ValueError, match="nesting"):
        CppScene(str(tmp_path / "deep.gltf"))

    # PNG headers with bit depths outside the table of the PNG specification (0 used to divide by zero, 3/5/6/7 read past the row)
    def png(depth, ctype, w=4, h=4):
        def chunk(t, b):
            return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
        raw = b"".join(b"\0" + bytes((w * max(depth, 1) * {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype] + 7) // 8) for _ in range(h))
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"PLTE", bytes(6)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    for depth, ctype in ((0, 0), (3, 0), (5, 3), (6, 0), (7, 3), (16, 3), (4, 2), (1, 6)):
        with pytest.raises(ValueError, match="unsupported colour type"):
            CppScene(_image_doc(tmp_path, png(depth, ctype), f"bad_{depth}_{ctype}.png"))
    ok = CppScene(_image_doc(tmp_path, png(2, 0), "ok_2_0.png"))
    assert ok.textures[0][0].shape == (4, 4, 4)


def test_png_trns_for_low_and_high_bit_depths(tmp_path):
    """tRNS of grey images below 8 bits and of 16-bit images (was honoured for 8 bits only)."""
    from PIL import Image
    g = Image.fromarray((np.arange(64, dtype=np.uint8).reshape(8, 8) % 4).astype(np.uint8), "L")
    b = io.BytesIO()
    g.point(lambda v: v * 85).convert("L").save(b, format="PNG", bits=2, transparency=85 * 2)
    cpp = CppScene(_image_doc(tmp_path, b.getvalue(), "trns2.png"))
    want = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGBA"))
    assert np.array_equal(cpp.textures[0][0], want)
    cpp.close()


def test_cpp_importer_sparse_accessors(tmp_path):
    from tests.test_gltf import sparse_doc
    path = _write(tmp_path, sparse_doc())
    cpp = CppScene(path)
    compare(gltf.load_gltf(path), cpp, exact=False)
    v = np.frombuffer(cpp.vertices.tobytes(), hd.vertex_dtype)
    assert np.array_equal(v["position"][2], [5, 6, 7]) and np.allclose(v["texcoord"][3], [0.75, 1.0], atol=1e-6)
    cpp.close()


def _image_doc(tmp_path, data, name):
    (tmp_path / name).write_bytes(data)
    doc = _tri_doc()
    doc["images"] = [{"uri": name}]
    doc["textures"] = [{"source": 0}]
    doc["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
    return _write(tmp_path, doc, name + ".gltf")


@pytest.mark.parametrize("mode", ["RGBA", "RGB", "L", "LA", "P"])
def test_png_decoder_against_pillow(tmp_path, mode):
    from PIL import Image
    rng = np.random.default_rng(5)
    src = Image.fromarray(rng.integers(0, 256, (13, 21, 4), dtype=np.uint8), "RGBA").convert(mode)
    b = io.BytesIO(); src.save(b, format="PNG")
    cpp = CppScene(_image_doc(tmp_path, b.getvalue(), f"img_{mode}.png"))
    assert np.array_equal(cpp.textures[0][0], np.asarray(src.convert("RGBA")))
    cpp.close()


def test_interlaced_png_against_pillow(tmp_path):
    """Adam7.  Pillow cannot write interlaced PNGs, so the file is assembled here: seven reduced images, filter 0, one zlib stream."""
    import struct, zlib
    from PIL import Image
    rng = np.random.default_rng(11)
    w, h = 19, 13
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    raw = b""
    for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = img[y0::dy, x0::dx]
        if sub.size:
            raw += b"".join(b"\x00" + sub[r].tobytes() for r in range(sub.shape[0]))
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")), img)       # Pillow reads it back as the source image
    cpp = CppScene(_image_doc(tmp_path, data, "adam7.png"))
    assert np.array_equal(cpp.textures[0][0], img)
    cpp.close()


@pytest.mark.parametrize("sub,gray,restart", [(0, False, 0), (2, False, 0), (0, True, 0), (1, False, 4)])
def test_jpeg_decoder_against_pillow(tmp_path, sub, gray, restart):
    """Baseline JPEG: within an LSB or two of libjpeg where no chroma upsampling is involved (decoders differ in IDCT rounding);
    with subsampled chroma libjpeg interpolates and this decoder replicates, so only the mean error is bounded there."""
    from PIL import Image
    y, x = np.mgrid[0:40, 0:56]
    img = np.stack([(x * 4) % 256, (y * 6) % 256, ((x + y) * 3) % 256], -1).astype(np.uint8)
    src = Image.fromarray(img, "RGB").convert("L") if gray else Image.fromarray(img, "RGB")
    b = io.BytesIO()
    kw = {} if gray else {"subsampling": sub}
    if restart:
        kw["restart_marker_rows"] = restart
    try:
        src.save(b, format="JPEG", quality=92, **kw)
    except TypeError:
        kw.pop("restart_marker_rows", None)
        src.save(b, format="JPEG", quality=92, **kw)
    ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGBA")).astype(int)
    cpp = CppScene(_image_doc(tmp_path, b.getvalue(), f"img_{sub}_{int(gray)}_{restart}.jpg"))
    got = cpp.textures[0][0].astype(int)
    cpp.close()
    assert got.shape == ref.shape and (got[..., 3] == 255).all()
    d = np.abs(got[..., :3] - ref[..., :3])
    if sub == 0 or gray:
        assert d.max() <= 3, d.max()
    else:
        assert d.mean() < 4.0, d.mean()


@pytest.mark.parametrize("sub", [0, 2])
def test_progressive_jpeg_against_pillow(tmp_path, sub):
    from PIL import Image
    rng = np.random.default_rng(9)
    y, x = np.mgrid[0:45, 0:70]
    img = np.stack([(x * 3 + rng.integers(0, 9, x.shape)) % 256, (y * 5) % 256, ((x * y) // 7) % 256], -1).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(img, "RGB").save(b, format="JPEG", quality=88, progressive=True, subsampling=sub)
    assert b"\xff\xc2" in b.getvalue()                      # really SOF2
    ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGBA")).astype(int)
    cpp = CppScene(_image_doc(tmp_path, b.getvalue(), f"prog_{sub}.jpg"))
    got = cpp.textures[0][0].astype(int)
    cpp.close()
    d = np.abs(got[..., :3] - ref[..., :3])
    if sub == 0:
        assert d.max() <= 3, d.max()
    else:
        assert d.mean() < 4.0, d.mean()


def test_arithmetic_or_garbage_jpeg_is_rejected_with_a_message(tmp_path):
    data = b"\xff\xd8\xff\xc9\x00\x0b\x08\x00\x10\x00\x10\x01\x01\x11\x00\xff\xd9"      # SOF9: arithmetic coding
    with pytest.raises(ValueError, match="not supported"):
        CppScene(_image_doc(tmp_path, data, "arith.jpg"))


def test_asset_dir_loads_through_the_cpp_importer(tmp_path, monkeypatch):
    """PT_ASSET_DIR (real assets are used when present): workloads.c3_sponza picks <dir>/Sponza.glb up through pt_gltf_load --
    the importer a C++ host uses -- and the flat arrays are those of the scene that was exported."""
    from vk_raytrace_amd import workloads
    from vk_raytrace_amd.scene import GltfFileScene
    src = synth.sponza_like(target_tris=3000, tex_size=8)
    gltf.save_gltf(src, str(tmp_path / "Sponza.glb"))
    monkeypatch.setenv("PT_ASSET_DIR", str(tmp_path))
    wl = workloads.c3_sponza(64, 48, 2)
    assert isinstance(wl.scene, GltfFileScene) and wl.note == "gltf" and "Sponza.glb" in wl.name
    src.finalize(capi.pack_vertices)
    d, _ = wl.scene.desc()
    assert d.numVertices == len(src.vertices) and d.numIndices == len(src.indices)
    got = np.ctypeslib.as_array(C.cast(d.vertices, C.POINTER(C.c_uint8)), (d.numVertices * 32,))
    assert np.array_equal(got, src.vertices.view(np.uint8).reshape(-1))
    assert wl.scene.num_triangles == src.num_triangles and len(wl.scene.materials) == len(src.materials) and len(wl.scene.textures) == len(src.textures)
    assert np.allclose(wl.scene.camera.eye, src.camera.eye, atol=1e-6) and abs(wl.scene.camera.fov - src.camera.fov) < 1e-4
    monkeypatch.setenv("PT_ASSET_IMPORTER", "python")
    assert not isinstance(workloads.c3_sponza(64, 48, 2).scene, GltfFileScene)
