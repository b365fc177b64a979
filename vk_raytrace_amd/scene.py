"""Host-side scene container: the flat arrays the reference's ``Scene`` uploads.

Mirrors the *output* of ``Scene::load`` (reference: src/scene.cpp:56-118): one
packed ``VertexAttributes[]`` + ``uint32 indices[]`` pool addressed per
prim-mesh (src/scene.cpp:190-274), ``GltfShadeMaterial[]`` (:339-382),
``Light[]`` (:304-333), RGBA8 textures with sampler state (:488-580), one node
per TLAS instance (src/accelstruct.cpp:137-159) and the camera
(src/scene.cpp:629-640).  Geometry arrives as raw attribute arrays and is packed
by ``pt_pack_vertices`` in libptmi.so (no GPU needed for that).
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import host_device as hd


@dataclass
class Texture:
    rgba8: np.ndarray  # (H, W, 4) uint8
    magFilter: int = hd.FILTER_LINEAR
    minFilter: int = hd.FILTER_LINEAR
    wrapS: int = hd.WRAP_REPEAT
    wrapT: int = hd.WRAP_REPEAT


@dataclass
class Camera:
    eye: tuple = (0.0, 0.0, 3.0)
    center: tuple = (0.0, 0.0, 0.0)
    up: tuple = (0.0, 1.0, 0.0)
    fov: float = 60.0  # degrees; nvh::CameraManipulator default
    aperture: float = 0.0
    focal_dist: Optional[float] = None  # None: |center - eye| like src/scene.cpp:639


class Scene:
    def __init__(self, name="scene"):
        self.name = name
        self._pos: List[np.ndarray] = []
        self._nrm: List[np.ndarray] = []
        self._tan: List[np.ndarray] = []
        self._uv: List[np.ndarray] = []
        self._col: List[np.ndarray] = []
        self._idx: List[np.ndarray] = []
        self._nverts = 0
        self._nidx = 0
        self.prim_meshes = []  # (vertexOffset, vertexCount, firstIndex, indexCount, materialIndex)
        self.nodes = []  # (4x4 row-major numpy world matrix, primMesh)
        self.materials: List[np.ndarray] = []
        self.lights: List[np.ndarray] = []
        self.textures: List[Texture] = []
        self.camera = Camera()
        # filled by finalize()
        self.vertices = None
        self.indices = None

    # ------------------------------------------------------------------ building
    def add_texture(self, rgba8, **kw) -> int:
        rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert rgba8.ndim == 3 and rgba8.shape[2] == 4
        self.textures.append(Texture(rgba8, **kw))
        return len(self.textures) - 1

    def add_material(self, **kw) -> int:
        m = hd.default_material()
        for k, v in kw.items():
            m[k] = v
        self.materials.append(m)
        return len(self.materials) - 1

    def add_light(self, **kw) -> int:
        l = np.zeros((), dtype=hd.light_dtype)
        l["color"] = (1, 1, 1)
        l["intensity"] = 1.0
        l["innerConeCos"] = 1.0
        l["outerConeCos"] = np.cos(np.pi / 4)
        for k, v in kw.items():
            l[k] = v
        self.lights.append(l)
        return len(self.lights) - 1

    def add_prim_mesh(self, positions, normals, uvs, indices, material, tangents=None, colors=None) -> int:
        """One glTF primitive (== one BLAS in the reference).  indices are relative to this mesh."""
        positions = np.asarray(positions, np.float32).reshape(-1, 3)
        n = len(positions)
        normals = np.asarray(normals, np.float32).reshape(n, 3)
        uvs = np.asarray(uvs, np.float32).reshape(n, 2)
        if tangents is None:
            tangents = default_tangents(normals)
        tangents = np.asarray(tangents, np.float32).reshape(n, 4)
        if colors is None:
            colors = np.ones((n, 4), np.float32)
        colors = np.asarray(colors, np.float32).reshape(n, 4)
        indices = np.asarray(indices, np.uint32).reshape(-1)
        assert len(indices) % 3 == 0 and (len(indices) == 0 or indices.max() < n)
        self._pos.append(positions); self._nrm.append(normals); self._tan.append(tangents)
        self._uv.append(uvs); self._col.append(colors); self._idx.append(indices)
        self.prim_meshes.append((self._nverts, n, self._nidx, len(indices), material))
        self._nverts += n
        self._nidx += len(indices)
        return len(self.prim_meshes) - 1

    def add_node(self, prim_mesh, matrix=None) -> int:
        m = np.eye(4, dtype=np.float32) if matrix is None else np.asarray(matrix, np.float32).reshape(4, 4)
        self.nodes.append((m, prim_mesh))
        return len(self.nodes) - 1

    # ------------------------------------------------------------------ packing
    def raw_attributes(self):
        cat = lambda l, w: (np.concatenate(l) if l else np.zeros((0, w), np.float32))
        return cat(self._pos, 3), cat(self._nrm, 3), cat(self._tan, 4), cat(self._uv, 2), cat(self._col, 4)

    def finalize(self, pack_fn):
        """pack_fn(pos, nrm, tan4, uv, col4) -> VertexAttributes[n]  (pt_pack_vertices)."""
        pos, nrm, tan, uv, col = self.raw_attributes()
        self.vertices = pack_fn(pos, nrm, tan, uv, col)
        self.indices = np.concatenate(self._idx).astype(np.uint32) if self._idx else np.zeros(0, np.uint32)
        return self

    @property
    def num_triangles(self):
        return sum(self.prim_meshes[pm][3] // 3 for _, pm in self.nodes)

    def node_array(self):
        """pt_Node[]: one TLAS instance per node (pt_set_scene, pt_update_instances)"""
        nd = np.zeros(len(self.nodes), hd.node_dtype)
        for i, (m, p) in enumerate(self.nodes):
            nd[i]["worldMatrix"] = m.T.reshape(16)  # row-major numpy -> column-major
            nd[i]["primMesh"] = p
        return nd

    def desc(self):
        """Builds the pt_SceneDesc; returns (desc, keepalive)."""
        assert self.vertices is not None, "call finalize() first"
        pm = np.zeros(len(self.prim_meshes), hd.primmesh_dtype)
        for i, t in enumerate(self.prim_meshes):
            pm[i] = t
        nd = self.node_array()
        mats = np.array(self.materials, dtype=hd.material_dtype) if self.materials else np.zeros(0, hd.material_dtype)
        lights = np.array(self.lights, dtype=hd.light_dtype) if self.lights else np.zeros(0, hd.light_dtype)
        tex = (hd.TextureDesc * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            tex[i] = hd.TextureDesc(t.rgba8.ctypes.data, t.rgba8.shape[1], t.rgba8.shape[0], t.magFilter, t.minFilter, t.wrapS, t.wrapT)
        verts = np.ascontiguousarray(self.vertices)
        idx = np.ascontiguousarray(self.indices)
        d = hd.SceneDesc(verts.ctypes.data, len(verts), idx.ctypes.data, len(idx), pm.ctypes.data, len(pm), nd.ctypes.data, len(nd),
                         mats.ctypes.data, len(mats), lights.ctypes.data if len(lights) else None, len(lights),
                         C.cast(tex, C.c_void_p).value if self.textures else None, len(self.textures))
        return d, (verts, idx, pm, nd, mats, lights, tex)


def default_tangents(normals):
    """A deterministic tangent frame for meshes that come without one."""
    n = np.asarray(normals, np.float32)
    ref = np.where(np.abs(n[:, 1:2]) < 0.99, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    t = np.cross(ref, n)
    t /= np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-20)
    return np.concatenate([t, np.ones((len(n), 1), np.float32)], axis=1).astype(np.float32)


def translate(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m


def scale(x, y=None, z=None):
    y = x if y is None else y
    z = x if z is None else z
    return np.diag(np.array([x, y, z, 1], np.float32))


def rotate_y(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def rotate_x(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4, dtype=np.float32)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def rotate_z(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


class GltfFileScene:
    """A .gltf / .glb file imported by libptmi's own C++ importer (pt_gltf_load: the Scene::load of the drop-in, reference
    src/scene.cpp:56-155) -- what a C++ host gets.  Same surface as `Scene` as far as the renderer, the oracle binding and bench.py use it
    (desc(), camera, counts, node_array()); the flat arrays stay owned by the importer until close()."""

    def __init__(self, path):
        from . import capi
        self.path = str(path)
        self.name = self.path
        self._L = capi.lib()
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self._L.pt_gltf_load(self.path.encode(), C.byref(self._h), err, 512)
        if rc != capi.PT_OK:
            raise ValueError(f"pt_gltf_load({self.path}): {err.value.decode()}")
        d = self._L.pt_gltf_desc(self._h).contents
        self._desc = d
        self.vertices = True  # (Scene.finalize() has nothing to do: the importer packs the vertices itself)
        self.prim_meshes = [tuple(int(x) for x in t) for t in
                            np.frombuffer(C.string_at(d.primMeshes, d.numPrimMeshes * hd.primmesh_dtype.itemsize), hd.primmesh_dtype).tolist()]
        self._nodes = np.frombuffer(C.string_at(d.nodes, d.numNodes * hd.node_dtype.itemsize), hd.node_dtype).copy()
        self.nodes = [(np.asarray(n["worldMatrix"], np.float32).reshape(4, 4).T.copy(), int(n["primMesh"])) for n in self._nodes]
        self.materials = list(np.frombuffer(C.string_at(d.materials, d.numMaterials * hd.material_dtype.itemsize), hd.material_dtype).copy())
        self.lights = list(np.frombuffer(C.string_at(d.lights, d.numLights * hd.light_dtype.itemsize), hd.light_dtype).copy()) if d.numLights else []
        self.textures = [None] * int(d.numTextures)
        e, c, u = (np.zeros(3, np.float32) for _ in range(3))
        f = C.c_float()
        self._L.pt_gltf_camera(self._h, e.ctypes.data, c.ctypes.data, u.ctypes.data, C.byref(f))
        self.camera = Camera(tuple(float(x) for x in e), tuple(float(x) for x in c), tuple(float(x) for x in u), float(f.value))

    def finalize(self, pack_fn=None):
        return self

    @property
    def num_triangles(self):
        return sum(self.prim_meshes[pm][3] // 3 for _, pm in self.nodes)

    def node_array(self):
        return self._nodes.copy()

    def desc(self):
        return self._desc, (self,)

    def close(self):
        if self._h:
            self._L.pt_gltf_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
