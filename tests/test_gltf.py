"""glTF import / export (vk_raytrace_amd/gltf.py): round trips of scenes that use every feature the renderer consumes
must give back the identical flat arrays; hand-written files cover what the exporter never emits (TRS nodes, shared
meshes, strided / normalised accessors, data uris, missing attributes, missing sampler, GLB)."""
import base64
import json
import math
import os
import struct

import numpy as np
import pytest

from vk_raytrace_amd import capi, gltf, host_device as hd, synth

EXACT_MAT_FIELDS = [n for n in hd.material_dtype.names if n not in ("anisotropyDirection", "_pad0", "_pad1")]


def assert_same_scene(a, b, exact_camera=False):
    a.finalize(capi.pack_vertices); b.finalize(capi.pack_vertices)
    assert np.array_equal(a.vertices.view(np.uint8), b.vertices.view(np.uint8))      # packed VertexAttributes, bit for bit
    assert np.array_equal(a.indices, b.indices)
    assert a.prim_meshes == b.prim_meshes
    assert len(a.nodes) == len(b.nodes)
    for (ma, pa), (mb, pb) in zip(a.nodes, b.nodes):
        assert pa == pb and np.array_equal(ma, mb)
    assert len(a.materials) == len(b.materials)
    for x, y in zip(a.materials, b.materials):
        for f in EXACT_MAT_FIELDS:
            assert np.array_equal(np.asarray(x[f]), np.asarray(y[f])), f
        assert np.allclose(x["anisotropyDirection"], y["anisotropyDirection"], atol=1e-6)
    assert len(a.textures) == len(b.textures)
    for x, y in zip(a.textures, b.textures):
        assert np.array_equal(x.rgba8, y.rgba8) and (x.magFilter, x.minFilter, x.wrapS, x.wrapT) == (y.magFilter, y.minFilter, y.wrapS, y.wrapT)
    assert len(a.lights) == len(b.lights)
    for x, y in zip(a.lights, b.lights):
        for f in ("color", "intensity", "range", "type", "position"):
            assert np.allclose(x[f], y[f], rtol=1e-6, atol=1e-6), f
        assert np.allclose(x["direction"] / np.linalg.norm(x["direction"]), y["direction"] / np.linalg.norm(y["direction"]), atol=1e-6)
        assert np.allclose([x["innerConeCos"], x["outerConeCos"]], [y["innerConeCos"], y["outerConeCos"]], atol=1e-6)
    assert np.allclose(a.camera.eye, b.camera.eye, atol=1e-6) and np.allclose(a.camera.center, b.camera.center, atol=1e-5)
    assert abs(a.camera.fov - b.camera.fov) < 1e-4


@pytest.mark.parametrize("ext", ["gltf", "glb"])
@pytest.mark.parametrize("make", [lambda: synth.feature_box(tex_size=16), lambda: synth.quad_scene(), lambda: synth.sponza_like(target_tris=3000, tex_size=8)])
def test_round_trip(tmp_path, make, ext):
    sc = make()
    path = str(tmp_path / f"scene.{ext}")
    gltf.save_gltf(sc, path)
    back = gltf.load_gltf(path)
    assert_same_scene(make(), back)


def _write(tmp_path, doc, name="t.gltf"):
    p = tmp_path / name
    p.write_text(json.dumps(doc))
    return str(p)


def _tri_doc(extra_attrs=None, **node):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    idx = np.array([0, 1, 2, 2, 1, 3], np.uint16)
    blob = pos.tobytes() + idx.tobytes()
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 12}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"}, {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}],
           "nodes": [dict({"mesh": 0}, **node)], "scenes": [{"nodes": [0]}], "scene": 0}
    return doc


def test_missing_attributes_are_synthesised(tmp_path):
    sc = gltf.load_gltf(_write(tmp_path, _tri_doc()))
    assert len(sc.materials) == 1 and int(sc.materials[0]["pbrBaseColorTexture"]) == -1      # default material
    pos, nrm, tan, uv, col = sc.raw_attributes()
    assert np.allclose(nrm, [[0, 0, 1]] * 4)                                                   # geometric normal of the quad
    assert np.allclose(np.abs((tan[:, :3] * nrm).sum(1)), 0, atol=1e-6) and np.allclose(np.linalg.norm(tan[:, :3], axis=1), 1)
    assert np.array_equal(uv, np.zeros((4, 2), np.float32)) and np.array_equal(col, np.ones((4, 4), np.float32))
    assert np.array_equal(sc._idx[0], [0, 1, 2, 2, 1, 3])
    # no camera in the file: the scene is framed
    assert sc.camera.center == pytest.approx((0.5, 0.5, 0.0)) and sc.camera.eye[2] > 1.0


def test_trs_nodes_children_and_shared_meshes(tmp_path):
    doc = _tri_doc()
    h = math.sqrt(0.5)
    doc["nodes"] = [{"children": [1, 2], "translation": [1, 2, 3]},
                    {"mesh": 0, "rotation": [0, 0, h, h], "scale": [2, 2, 2]},          # 90 degrees about z, then the parent's translation
                    {"mesh": 0}]
    doc["scenes"] = [{"nodes": [0]}]
    sc = gltf.load_gltf(_write(tmp_path, doc))
    assert len(sc.prim_meshes) == 1 and [pm for _, pm in sc.nodes] == [0, 0]            # one BLAS, two instances
    m = sc.nodes[0][0]
    assert np.allclose(m @ np.array([1, 0, 0, 1]), [1, 4, 3, 1], atol=1e-6)            # x -> +y (scaled by 2), then translated
    assert np.allclose(sc.nodes[1][0], [[1, 0, 0, 1], [0, 1, 0, 2], [0, 0, 1, 3], [0, 0, 0, 1]])


def test_strided_normalised_accessors_and_texture_transform(tmp_path):
    # interleaved: position (3 f32) + uv as normalised u16 (2 x u16) -> stride 16
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    uvq = np.array([[0, 0], [65535, 0], [0, 32768]], np.uint16)
    inter = b"".join(pos[i].tobytes() + uvq[i].tobytes() for i in range(3))
    img = np.zeros((2, 2, 4), np.uint8); img[..., 0] = [[10, 20], [30, 40]]; img[..., 3] = 255
    import io
    from PIL import Image
    b = io.BytesIO(); Image.fromarray(img, "RGBA").save(b, format="PNG")
    blob = inter + b.getvalue()
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48, "byteStride": 16}, {"buffer": 0, "byteOffset": 48, "byteLength": len(b.getvalue())}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                         {"bufferView": 0, "byteOffset": 12, "componentType": 5123, "normalized": True, "count": 3, "type": "VEC2"}],
           "images": [{"bufferView": 1, "mimeType": "image/png"}], "textures": [{"source": 0}],
           "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0, "extensions": {"KHR_texture_transform": {"offset": [0.25, 0.5], "scale": [2, 3], "rotation": 0.0}}}},
                          "alphaMode": "MASK", "extensions": {"KHR_materials_ior": {"ior": 1.33}, "KHR_materials_sheen": {"sheenColorFactor": [1, 0.5, 0], "sheenRoughnessFactor": 0.2}}}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "material": 0}]}],     # no indices: 0..n-1
           "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}]}
    sc = gltf.load_gltf(_write(tmp_path, doc))
    assert np.allclose(sc._uv[0], [[0, 0], [1, 0], [0, 32768 / 65535]])
    assert np.array_equal(sc._idx[0], [0, 1, 2])
    t = sc.textures[0]
    assert np.array_equal(t.rgba8, img) and (t.magFilter, t.wrapS) == (hd.FILTER_LINEAR, hd.WRAP_REPEAT)   # no sampler: LINEAR / REPEAT
    m = sc.materials[0]
    M = np.asarray(m["uvTransform"]).reshape(4, 4)              # column-major: M[c] is column c
    u, v = 0.5, 0.25
    row = np.array([u, v, 1, 1], np.float32)
    assert np.allclose([row @ M[0], row @ M[1]], [2 * u + 0.25, 3 * v + 0.5])   # the shader's (uv, 1, 1) * uvTransform
    assert int(m["alphaMode"]) == hd.ALPHA_MASK and float(m["ior"]) == np.float32(1.33)
    assert int(m["sheen"]) == (255 | (128 << 8) | (0 << 16) | (51 << 24))


def test_rejects_what_it_cannot_represent(tmp_path):
    doc = _tri_doc()
    doc["accessors"][0]["sparse"] = {"count": 1, "indices": {"bufferView": 1, "componentType": 5123}, "values": {"bufferView": 0}}
    doc["accessors"][0]["sparse"]["indices"]["byteOffset"] = 10      # index 3 (u16 at offset 10 of the index view) is fine ...
    gltf.load_gltf(_write(tmp_path, doc))
    doc["accessors"][0]["count"] = 3                                  # ... but not for a 3-element accessor
    doc["meshes"][0]["primitives"][0].pop("indices")
    with pytest.raises(gltf.GltfError):
        gltf.load_gltf(_write(tmp_path, doc))
    doc = _tri_doc()
    doc["asset"]["version"] = "1.0"
    with pytest.raises(gltf.GltfError):
        gltf.load_gltf(_write(tmp_path, doc))
    doc = _tri_doc()
    doc["meshes"][0]["primitives"][0]["mode"] = 1           # lines: not drawable, silently skipped like the reference
    assert gltf.load_gltf(_write(tmp_path, doc)).nodes == []


def sparse_doc():
    """The quad of _tri_doc with vertex 2 moved by a sparse accessor, and an accessor without bufferView (zeros) + sparse values as uv."""
    import struct
    doc = _tri_doc()
    raw = base64.b64decode(doc["buffers"][0]["uri"].split(",")[1])
    extra = struct.pack("<I", 2) + struct.pack("<3f", 5.0, 6.0, 7.0) + struct.pack("<HH", 1, 3) + struct.pack("<4f", 0.25, 0.5, 0.75, 1.0)
    blob = raw + extra
    doc["buffers"][0] = {"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}
    doc["bufferViews"] += [{"buffer": 0, "byteOffset": 60, "byteLength": 4}, {"buffer": 0, "byteOffset": 64, "byteLength": 12},
                           {"buffer": 0, "byteOffset": 76, "byteLength": 4}, {"buffer": 0, "byteOffset": 80, "byteLength": 16}]
    doc["accessors"][0]["sparse"] = {"count": 1, "indices": {"bufferView": 2, "componentType": 5125}, "values": {"bufferView": 3}}
    doc["accessors"].append({"componentType": 5126, "count": 4, "type": "VEC2",
                             "sparse": {"count": 2, "indices": {"bufferView": 4, "componentType": 5123}, "values": {"bufferView": 5}}})
    doc["meshes"][0]["primitives"][0]["attributes"]["TEXCOORD_0"] = 2
    return doc


def test_sparse_accessors(tmp_path):
    sc = gltf.load_gltf(_write(tmp_path, sparse_doc()))
    assert np.array_equal(sc._pos[0], [[0, 0, 0], [1, 0, 0], [5, 6, 7], [1, 1, 0]])
    assert np.array_equal(sc._uv[0], [[0, 0], [0.25, 0.5], [0, 0], [0.75, 1.0]])

