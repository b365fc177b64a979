"""One hand-written glTF fixture per import rule that the reference leaves to the un-vendored nvh::GltfScene
(reference: src/scene.cpp:56-76 -- tinygltf parse, `m_gltf.importMaterials`, `m_gltf.importDrawableNodes(Normal | Texcoord_0 | Tangent | Color_0)`;
nvpro_core is a git submodule that is absent from /root/reference, INTEGRATION.md section 2 lists the rules).  Every rule is a CHOICE made here and
is therefore stated, next to the fixture that would have to change if a maintainer with nvpro_core finds the library does otherwise.  Each
fixture goes through BOTH importers -- vk_raytrace_amd/gltf.py and libptmi's pt_gltf_load (csrc/pt_gltf.cpp) -- which must agree.
No GPU involved."""
import base64
import json
import math

import numpy as np
import pytest

from vk_raytrace_amd import capi, gltf, host_device as hd
from tests.test_gltf_cpp import CppScene, compare


def _doc(arrays, attributes, indices=None, material=None, nodes=None, **top):
    """arrays: list of (numpy array, gltf type, componentType, normalized); attributes: name -> array index"""
    blob, views, accs = b"", [], []
    for a, typ, ct, norm in arrays:
        while len(blob) % 4:
            blob += b"\0"
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": a.nbytes})
        acc = {"bufferView": len(views) - 1, "componentType": ct, "count": len(a), "type": typ}
        if norm:
            acc["normalized"] = True
        accs.append(acc)
        blob += a.tobytes()
    prim = {"attributes": attributes}
    if indices is not None:
        prim["indices"] = indices
    if material is not None:
        prim["material"] = material
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
           "bufferViews": views, "accessors": accs, "meshes": [{"primitives": [prim]}], "nodes": nodes or [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0}
    doc.update(top)
    return doc


def _load_both(tmp_path, doc, name="rule.gltf"):
    p = tmp_path / name
    p.write_text(json.dumps(doc))
    py, cpp = gltf.load_gltf(str(p)), CppScene(str(p))
    compare(py, cpp, exact=False)
    a = np.frombuffer(cpp.vertices.tobytes(), hd.vertex_dtype)
    py.finalize(capi.pack_vertices)
    # the packed attributes (octahedral normal / tangent, uv with the handedness bit, RGBA8 colour: src/scene.cpp:219-242) agree bit for bit
    for f in ("normal", "tangent", "color"):
        assert np.array_equal(a[f], py.vertices[f]), f
    cpp.close()
    return py


ROOF_POS = np.array([[0, 0, 0], [1, 0, 0], [0.5, 0.5, 1], [0, 1, 0], [1, 1, 0]], np.float32)  # two slopes meeting at a ridge vertex (2)
ROOF_IDX = np.array([0, 1, 2, 3, 2, 4], np.uint16)                                              # front slope, back slope
F32, U16, U8 = 5126, 5123, 5121


def test_rule_missing_normal_is_smooth_area_weighted(tmp_path):
    """RULE: no NORMAL accessor -> per-vertex normals = normalised sum of the (unnormalised, i.e. area-weighted) face normals of the primitive's
    triangles that use the vertex -- smooth, NOT one flat normal per face (a flat rule would have to split the shared ridge vertex).
    (glTF 2.0 section 3.7.2.1 asks for flat normals; nvh::GltfScene's choice is unpinned -- if it is flat, this fixture changes.)"""
    py = _load_both(tmp_path, _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False)], {"POSITION": 0}, indices=1))
    pos, nrm, tan, uv, col = py.raw_attributes()
    n_front = np.cross(ROOF_POS[1] - ROOF_POS[0], ROOF_POS[2] - ROOF_POS[0])
    n_back = np.cross(ROOF_POS[2] - ROOF_POS[3], ROOF_POS[4] - ROOF_POS[3])
    assert np.allclose(nrm[0], n_front / np.linalg.norm(n_front), atol=1e-6)         # used by the front face only
    ridge = n_front + n_back                                                         # area-weighted: the cross products are not normalised first
    assert np.allclose(nrm[2], ridge / np.linalg.norm(ridge), atol=1e-6)
    assert len(pos) == 5                                                             # no vertex was split


def test_rule_missing_tangent_comes_from_the_uv_parameterisation(tmp_path):
    """RULE: no TANGENT accessor but TEXCOORD_0 present -> per-vertex tangent = direction of increasing u (summed over the vertex's triangles),
    Gram-Schmidt against the normal; w = handedness = sign of dot(cross(N, T), direction of increasing v) (stored in the uv's LSB, scene.cpp:233)."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    nrm = np.array([[0, 0, 1]] * 4, np.float32)
    idx = np.array([0, 1, 2, 2, 1, 3], np.uint16)
    for uv, want_t, want_w in ((np.array([[0, 0], [1, 0], [0, 1], [1, 1]], np.float32), [1, 0, 0], 1.0),      # u along +x, v along +y: right-handed
                               (np.array([[0, 0], [0, 1], [1, 0], [1, 1]], np.float32), [0, 1, 0], -1.0)):    # u along +y, v along +x: mirrored
        py = _load_both(tmp_path, _doc([(pos, "VEC3", F32, False), (nrm, "VEC3", F32, False), (uv, "VEC2", F32, False), (idx, "SCALAR", U16, False)],
                                       {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, indices=3))
        tan = py.raw_attributes()[2]
        assert np.allclose(tan[:, :3], [want_t] * 4, atol=1e-6) and np.allclose(tan[:, 3], want_w)


def test_rule_missing_tangent_without_usable_uv_is_any_orthogonal_unit_vector(tmp_path):
    """RULE: no TANGENT and a degenerate uv parameterisation (here: no TEXCOORD_0 at all, so every uv is (0, 0)) -> a unit vector orthogonal to
    the normal (the frame of shaders/common.glsl:80-92), handedness +1."""
    py = _load_both(tmp_path, _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False)], {"POSITION": 0}, indices=1))
    pos, nrm, tan, uv, col = py.raw_attributes()
    assert np.allclose(np.linalg.norm(tan[:, :3], axis=1), 1, atol=1e-6) and np.allclose((tan[:, :3] * nrm).sum(1), 0, atol=1e-6)
    assert np.allclose(tan[:, 3], 1.0)


def test_rule_missing_texcoord_and_colour_defaults(tmp_path):
    """RULE: no TEXCOORD_0 -> (0, 0) for every vertex; no COLOR_0 -> (1, 1, 1, 1) (the shader multiplies albedo by the vertex colour,
    shaders/pathtrace.glsl:251)."""
    py = _load_both(tmp_path, _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False)], {"POSITION": 0}, indices=1))
    pos, nrm, tan, uv, col = py.raw_attributes()
    assert np.array_equal(uv, np.zeros((5, 2), np.float32)) and np.array_equal(col, np.ones((5, 4), np.float32))


def test_rule_colour_accessor_forms(tmp_path):
    """RULE: COLOR_0 may be VEC3 (alpha = 1) or VEC4, float or normalised u8 / u16 (glTF 2.0 section 3.7.2.1); the packed colour is RGBA8 with
    round-to-nearest (scene.cpp:238-241)."""
    c3 = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128], [10, 20, 30]], np.uint8)
    c3p = np.concatenate([c3, np.zeros((5, 1), np.uint8)], axis=1)   # VEC3 of u8 is padded to a 4-byte stride by the specification
    doc = _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False), (c3p, "VEC3", U8, True)], {"POSITION": 0, "COLOR_0": 2}, indices=1)
    doc["bufferViews"][2]["byteStride"] = 4
    py = _load_both(tmp_path, doc)
    col = py.raw_attributes()[4]
    assert np.allclose(col[:, :3], c3 / 255.0, atol=1e-7) and np.array_equal(col[:, 3], np.ones(5, np.float32))
    c4 = np.array([[65535, 0, 0, 32768]] * 5, np.uint16)
    py = _load_both(tmp_path, _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False), (c4, "VEC4", U16, True)], {"POSITION": 0, "COLOR_0": 2}, indices=1), "c4.gltf")
    col = py.raw_attributes()[4]
    assert np.allclose(col, [[1, 0, 0, 32768 / 65535]] * 5, atol=1e-7)


def test_rule_texture_transform_with_rotation(tmp_path):
    """RULE: KHR_texture_transform on the BASE-COLOUR texture becomes the material's one uvTransform (host_device.h:171; the shader applies it to
    every texture of the material, gltf_material.glsl:52-58): uv' = T(offset) * R(rotation) * S(scale) * uv as the extension defines it.
    Transforms on other textures of the material are ignored (the struct has nowhere to put them)."""
    import io
    from PIL import Image
    b = io.BytesIO(); Image.fromarray(np.full((2, 2, 4), 255, np.uint8), "RGBA").save(b, format="PNG")
    uvs = np.array([[0, 0], [1, 0], [0.5, 1], [0, 1], [1, 1]], np.float32)
    rot, off, sc = 0.5, [0.1, 0.2], [2.0, 3.0]
    doc = _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False), (uvs, "VEC2", F32, False), (np.frombuffer(b.getvalue(), np.uint8), "SCALAR", U8, False)],
               {"POSITION": 0, "TEXCOORD_0": 2}, indices=1, material=0,
               images=[{"bufferView": 3, "mimeType": "image/png"}], textures=[{"source": 0}],
               materials=[{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0, "extensions": {"KHR_texture_transform": {"offset": off, "scale": sc, "rotation": rot}}},
                                                   "metallicRoughnessTexture": {"index": 0, "extensions": {"KHR_texture_transform": {"offset": [9, 9]}}}}}])
    del doc["accessors"][3]   # the image lives in a buffer view, not an accessor
    py = _load_both(tmp_path, doc)
    M = np.asarray(py.materials[0]["uvTransform"]).reshape(4, 4)
    for u, v in ((0.3, 0.7), (1.0, 0.0)):
        row = np.array([u, v, 1, 1], np.float32)
        c, s = math.cos(rot), math.sin(rot)
        want = (off[0] + c * sc[0] * u + s * sc[1] * v, off[1] - s * sc[0] * u + c * sc[1] * v)   # KHR_texture_transform: rotation matrix [[c, s], [-s, c]]
        assert np.allclose([row @ M[0], row @ M[1]], want, atol=1e-6)


def test_rule_material_defaults_and_extension_defaults(tmp_path):
    """RULE: a material without the optional members takes the glTF 2.0 defaults (base colour (1,1,1,1), metallic 1, roughness 1, emissive 0, alphaMode
    OPAQUE, alphaCutoff 0.5, single sided, every texture index -1), and the KHR_materials_* extensions their specified defaults when absent: ior 1.5,
    transmission 0, clearcoat 0 / roughness 0, sheen 0, anisotropy 0, thickness 0 (thin walled), attenuation colour (1,1,1), lit.
    GltfShadeMaterial (host_device.h:133-179) is filled field by field by src/scene.cpp:344-378 from nvh::GltfMaterial, whose defaults are the library's."""
    py = _load_both(tmp_path, _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False)], {"POSITION": 0}, indices=1, material=0, materials=[{"name": "bare"}]))
    m = py.materials[0]
    assert np.array_equal(m["pbrBaseColorFactor"], [1, 1, 1, 1]) and float(m["pbrMetallicFactor"]) == 1.0 and float(m["pbrRoughnessFactor"]) == 1.0
    assert np.array_equal(m["emissiveFactor"], [0, 0, 0]) and int(m["alphaMode"]) == hd.ALPHA_OPAQUE and float(m["alphaCutoff"]) == 0.5 and int(m["doubleSided"]) == 0
    for t in ("pbrBaseColorTexture", "pbrMetallicRoughnessTexture", "normalTexture", "emissiveTexture", "transmissionTexture", "clearcoatTexture", "clearcoatRoughnessTexture"):
        assert int(m[t]) == -1, t
    assert float(m["ior"]) == 1.5 and float(m["transmissionFactor"]) == 0.0 and float(m["clearcoatFactor"]) == 0.0 and float(m["clearcoatRoughness"]) == 0.0
    assert int(m["sheen"]) == 0 and float(m["anisotropy"]) == 0.0 and float(m["thicknessFactor"]) == 0.0 and int(m["unlit"]) == 0
    assert np.array_equal(m["attenuationColor"], [1, 1, 1]) and float(m["normalTextureScale"]) == 1.0
    M = np.asarray(m["uvTransform"]).reshape(4, 4)
    assert np.array_equal(M, np.eye(4, dtype=np.float32))


def test_rule_primitive_without_material_uses_material_zero(tmp_path):
    """RULE: a primitive without `material` shades with the file's material 0 -- the behaviour of `std::max(0, material)` (and the kernels clamp a
    negative index the same way, csrc/pt_shade.h `matIndex < 0 ? 0`); a default material is
    created only for a file that has NO material at all (the fixture of test_gltf.py::test_missing_attributes_are_synthesised).  glTF 2.0 asks for
    the default material instead; which of the two nvh::GltfScene does is unpinned -- this fixture is the one to flip."""
    doc = _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False)], {"POSITION": 0}, indices=1, materials=[{"pbrMetallicRoughness": {"metallicFactor": 0.25}}])
    py = _load_both(tmp_path, doc)
    assert len(py.materials) == 1 and float(py.materials[0]["pbrMetallicFactor"]) == 0.25
    assert int(py.prim_meshes[0][4]) == 0   # (vertexOffset, vertexCount, firstIndex, indexCount, materialIndex)
    del doc["materials"]
    py = _load_both(tmp_path, doc, "nomat.gltf")
    assert len(py.materials) == 1 and float(py.materials[0]["pbrMetallicFactor"]) == 1.0


def test_rule_one_instance_per_node_and_primitive(tmp_path):
    """RULE: a mesh with several primitives gives one prim-mesh per primitive and one instance per (node, primitive) -- importDrawableNodes flattens the
    scene graph (scene.cpp:64-66); the instance's custom index is its prim-mesh (accelstruct.cpp:131-150)."""
    doc = _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False), (ROOF_IDX[:3].copy(), "SCALAR", U16, False)], {"POSITION": 0}, indices=1)
    doc["meshes"][0]["primitives"].append({"attributes": {"POSITION": 0}, "indices": 2})
    doc["nodes"] = [{"mesh": 0}, {"mesh": 0, "translation": [5, 0, 0]}]
    doc["scenes"] = [{"nodes": [0, 1]}]
    py = _load_both(tmp_path, doc)
    assert len(py.prim_meshes) == 2 and sorted(pm for _, pm in py.nodes) == [0, 0, 1, 1]


def test_rule_punctual_light_defaults_and_orientation(tmp_path):
    """RULE (KHR_lights_punctual): colour (1,1,1), intensity 1, range 0 = unlimited (punctual.glsl:28-31), spot cone inner 0 / outer pi/4 when absent; the light
    sits at the node's world translation and shines along the node's -z axis (scene.cpp:304-342 copies nvh::GltfLight)."""
    doc = _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False)], {"POSITION": 0}, indices=1)
    h = math.sqrt(0.5)
    doc["nodes"] = [{"mesh": 0}, {"translation": [1, 2, 3], "rotation": [h, 0, 0, h], "extensions": {"KHR_lights_punctual": {"light": 0}}},   # +90 degrees about x: -z -> +y
                    {"extensions": {"KHR_lights_punctual": {"light": 1}}}]
    doc["scenes"] = [{"nodes": [0, 1, 2]}]
    doc["extensions"] = {"KHR_lights_punctual": {"lights": [{"type": "spot"}, {"type": "directional", "color": [1, 0.5, 0.25], "intensity": 3.0}]}}
    doc["extensionsUsed"] = ["KHR_lights_punctual"]
    py = _load_both(tmp_path, doc)
    assert len(py.lights) == 2
    s, d = py.lights
    assert int(s["type"]) == hd.LightType_Spot and np.allclose(s["position"], [1, 2, 3]) and np.allclose(s["direction"] / np.linalg.norm(s["direction"]), [0, 1, 0], atol=1e-6)
    assert np.allclose(s["color"], [1, 1, 1]) and float(s["intensity"]) == 1.0 and float(s["range"]) == 0.0
    assert abs(float(s["innerConeCos"]) - 1.0) < 1e-6 and abs(float(s["outerConeCos"]) - math.cos(math.pi / 4)) < 1e-6
    assert int(d["type"]) == hd.LightType_Directional and np.allclose(d["color"], [1, 0.5, 0.25]) and float(d["intensity"]) == 3.0
    assert np.allclose(d["direction"] / np.linalg.norm(d["direction"]), [0, 0, -1], atol=1e-6)


def test_rule_camera_first_perspective_camera_node(tmp_path):
    """RULE: the first camera node met in the scene-graph walk supplies eye (its translation), viewing direction (its -z) and fov = yfov in degrees
    (nvh::CameraManipulator is fed from the glTF camera in scene.cpp:286-297); no camera -> the scene's bounding box is framed (tested in test_gltf.py)."""
    doc = _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False)], {"POSITION": 0}, indices=1)
    doc["cameras"] = [{"type": "perspective", "perspective": {"yfov": math.radians(35.0), "znear": 0.1}}, {"type": "perspective", "perspective": {"yfov": 1.0, "znear": 0.1}}]
    doc["nodes"] = [{"mesh": 0}, {"camera": 0, "translation": [0.5, 0.5, 4.0]}, {"camera": 1, "translation": [9, 9, 9]}]
    doc["scenes"] = [{"nodes": [0, 1, 2]}]
    py = _load_both(tmp_path, doc)
    assert np.allclose(py.camera.eye, [0.5, 0.5, 4.0]) and abs(py.camera.fov - 35.0) < 1e-4
    fwd = np.asarray(py.camera.center) - np.asarray(py.camera.eye)
    assert np.allclose(fwd / np.linalg.norm(fwd), [0, 0, -1], atol=1e-6)


def test_rule_sampler_defaults(tmp_path):
    """RULE: a texture without a sampler is LINEAR / REPEAT (scene.cpp:561-571 builds that VkSamplerCreateInfo for it); a sampler without filters gives
    tinygltf's -1, which the reference's lookup table turns into enum 0 = NEAREST (scene.cpp:447-482), wrap modes default to REPEAT."""
    import io
    from PIL import Image
    b = io.BytesIO(); Image.fromarray(np.full((2, 2, 4), 200, np.uint8), "RGBA").save(b, format="PNG")
    uvs = np.zeros((5, 2), np.float32)
    doc = _doc([(ROOF_POS, "VEC3", F32, False), (ROOF_IDX, "SCALAR", U16, False), (uvs, "VEC2", F32, False), (np.frombuffer(b.getvalue(), np.uint8), "SCALAR", U8, False)],
               {"POSITION": 0, "TEXCOORD_0": 2}, indices=1, material=0, images=[{"bufferView": 3, "mimeType": "image/png"}],
               samplers=[{}, {"magFilter": 9729, "wrapS": 33071, "wrapT": 33648}], textures=[{"source": 0}, {"source": 0, "sampler": 0}, {"source": 0, "sampler": 1}],
               materials=[{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 1}}, "emissiveTexture": {"index": 2}}])
    del doc["accessors"][3]
    py = _load_both(tmp_path, doc)
    t0, t1, t2 = py.textures
    assert (t0.magFilter, t0.wrapS, t0.wrapT) == (hd.FILTER_LINEAR, hd.WRAP_REPEAT, hd.WRAP_REPEAT)
    assert (t1.magFilter, t1.wrapS, t1.wrapT) == (hd.FILTER_NEAREST, hd.WRAP_REPEAT, hd.WRAP_REPEAT)
    assert (t2.magFilter, t2.wrapS, t2.wrapT) == (hd.FILTER_LINEAR, hd.WRAP_CLAMP_TO_EDGE, hd.WRAP_MIRRORED_REPEAT)
