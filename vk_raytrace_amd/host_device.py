"""ctypes / numpy mirrors of ``include/pt_types.h``.

These are the structures the reference shares between host and device
(reference: shaders/host_device.h:107-281) plus the flat scene description that
replaces its descriptor sets (reference: shaders/layouts.glsl:37-52).  Field
names are the reference's.  ``np.dtype(Struct)`` gives the matching numpy record
type for array uploads.
"""
import ctypes as C

import numpy as np

# reference: shaders/host_device.h:88-102
eNoDebug, eBaseColor, eNormal, eMetallic, eEmissive, eAlpha, eRoughness, eTexcoord, eTangent, eRadiance, eWeight, eRayDir, eHeatmap = range(13)
ALPHA_OPAQUE, ALPHA_MASK, ALPHA_BLEND = 0, 1, 2  # host_device.h:126-131
LightType_Directional, LightType_Point, LightType_Spot = 0, 1, 2  # host_device.h:211-213
FILTER_NEAREST, FILTER_LINEAR = 0, 1
WRAP_REPEAT, WRAP_MIRRORED_REPEAT, WRAP_CLAMP_TO_EDGE = 0, 1, 2
TILE = 32  # PT_TILE


class RtxState(C.Structure):  # host_device.h:183-196
    _fields_ = [("frame", C.c_int32), ("maxDepth", C.c_int32), ("maxSamples", C.c_int32),
                ("fireflyClampThreshold", C.c_float), ("hdrMultiplier", C.c_float),
                ("debugging_mode", C.c_int32), ("pbrMode", C.c_int32), ("_pad0", C.c_int32),
                ("size", C.c_int32 * 2), ("minHeatmap", C.c_int32), ("maxHeatmap", C.c_int32)]


class SceneCamera(C.Structure):  # host_device.h:107-115
    _fields_ = [("viewInverse", C.c_float * 16), ("projInverse", C.c_float * 16),
                ("focalDist", C.c_float), ("aperture", C.c_float), ("nbLights", C.c_int32)]


class VertexAttributes(C.Structure):  # host_device.h:117-124
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_uint32), ("texcoord", C.c_float * 2),
                ("tangent", C.c_uint32), ("color", C.c_uint32)]


class GltfShadeMaterial(C.Structure):  # host_device.h:133-179
    _fields_ = [("pbrBaseColorFactor", C.c_float * 4), ("pbrBaseColorTexture", C.c_int32),
                ("pbrMetallicFactor", C.c_float), ("pbrRoughnessFactor", C.c_float),
                ("pbrMetallicRoughnessTexture", C.c_int32), ("emissiveTexture", C.c_int32), ("_pad0", C.c_int32),
                ("emissiveFactor", C.c_float * 3), ("alphaMode", C.c_int32), ("alphaCutoff", C.c_float),
                ("doubleSided", C.c_int32), ("normalTexture", C.c_int32), ("normalTextureScale", C.c_float),
                ("uvTransform", C.c_float * 16), ("unlit", C.c_int32), ("transmissionFactor", C.c_float),
                ("transmissionTexture", C.c_int32), ("ior", C.c_float), ("anisotropyDirection", C.c_float * 3),
                ("anisotropy", C.c_float), ("attenuationColor", C.c_float * 3), ("thicknessFactor", C.c_float),
                ("thicknessTexture", C.c_int32), ("attenuationDistance", C.c_float), ("clearcoatFactor", C.c_float),
                ("clearcoatRoughness", C.c_float), ("clearcoatTexture", C.c_int32),
                ("clearcoatRoughnessTexture", C.c_int32), ("sheen", C.c_uint32), ("_pad1", C.c_int32)]


class Light(C.Structure):  # host_device.h:215-230
    _fields_ = [("direction", C.c_float * 3), ("range", C.c_float), ("color", C.c_float * 3), ("intensity", C.c_float),
                ("position", C.c_float * 3), ("innerConeCos", C.c_float), ("outerConeCos", C.c_float),
                ("type", C.c_int32), ("padding", C.c_float * 2)]


class EnvAccel(C.Structure):  # host_device.h:233-239
    _fields_ = [("alias", C.c_uint32), ("q", C.c_float), ("pdf", C.c_float), ("aliasPdf", C.c_float)]


class Tonemapper(C.Structure):  # host_device.h:242-255
    _fields_ = [("brightness", C.c_float), ("contrast", C.c_float), ("saturation", C.c_float), ("vignette", C.c_float),
                ("avgLum", C.c_float), ("zoom", C.c_float), ("renderingRatio", C.c_float * 2),
                ("autoExposure", C.c_int32), ("Ywhite", C.c_float), ("key", C.c_float), ("dither", C.c_int32)]


class SunAndSky(C.Structure):  # host_device.h:258-281
    _fields_ = [("rgb_unit_conversion", C.c_float * 3), ("multiplier", C.c_float), ("haze", C.c_float),
                ("redblueshift", C.c_float), ("saturation", C.c_float), ("horizon_height", C.c_float),
                ("ground_color", C.c_float * 3), ("horizon_blur", C.c_float), ("night_color", C.c_float * 3),
                ("sun_disk_intensity", C.c_float), ("sun_direction", C.c_float * 3), ("sun_disk_scale", C.c_float),
                ("sun_glow_intensity", C.c_float), ("y_is_up", C.c_int32), ("physically_scaled_sun", C.c_int32),
                ("in_use", C.c_int32)]


class PrimMesh(C.Structure):  # nvh::GltfPrimMesh fields used at src/scene.cpp:205-252
    _fields_ = [("vertexOffset", C.c_uint32), ("vertexCount", C.c_uint32), ("firstIndex", C.c_uint32),
                ("indexCount", C.c_uint32), ("materialIndex", C.c_int32)]


class Node(C.Structure):  # nvh::GltfNode fields used at src/accelstruct.cpp:137-159
    _fields_ = [("worldMatrix", C.c_float * 16), ("primMesh", C.c_int32)]


class TextureDesc(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("magFilter", C.c_int32),
                ("minFilter", C.c_int32), ("wrapS", C.c_int32), ("wrapT", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("numVertices", C.c_uint32),
                ("indices", C.c_void_p), ("numIndices", C.c_uint32),
                ("primMeshes", C.c_void_p), ("numPrimMeshes", C.c_uint32),
                ("nodes", C.c_void_p), ("numNodes", C.c_uint32),
                ("materials", C.c_void_p), ("numMaterials", C.c_uint32),
                ("lights", C.c_void_p), ("numLights", C.c_uint32),
                ("textures", C.c_void_p), ("numTextures", C.c_uint32)]


class PickResult(C.Structure):
    _fields_ = [("worldRayOrigin", C.c_float * 3), ("hitT", C.c_float), ("worldRayDirection", C.c_float * 3), ("primitiveID", C.c_int32),
                ("instanceID", C.c_uint32), ("instanceCustomIndex", C.c_int32), ("baryCoord", C.c_float * 3)]


class Peaks(C.Structure):
    """pt_Peaks (pt_measure_peaks): ceilings measured on the device"""
    _fields_ = [("valuWaveInstrPerSec", C.c_double), ("hbmCopyBytesPerSec", C.c_double), ("hbmReadBytesPerSec", C.c_double), ("computeUnits", C.c_int32),
                ("clockMHz", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closestRays", C.c_uint64), ("shadowRays", C.c_uint64),
                ("shadedHits", C.c_uint64), ("misses", C.c_uint64), ("alphaTests", C.c_uint64),
                ("neeLookups", C.c_uint64), ("nodesVisited", C.c_uint64), ("trisTested", C.c_uint64),
                ("msGenerate", C.c_double), ("msTraceClosest", C.c_double), ("msShade", C.c_double),
                ("msTraceShadow", C.c_double), ("msAccumulate", C.c_double), ("launchesTraceClosest", C.c_uint64),
                ("numTriangles", C.c_uint32), ("numBvhNodes", C.c_uint32), ("msBuildAccel", C.c_double),
                ("bytesScene", C.c_uint64), ("bytesAccel", C.c_uint64), ("numBlas", C.c_uint32), ("numTlasNodes", C.c_uint32),
                ("msTail", C.c_double), ("batchFrames", C.c_uint32), ("framesInFlight", C.c_uint32),
                ("tailClosestRays", C.c_uint64), ("tailShadowRays", C.c_uint64), ("tailShadedHits", C.c_uint64), ("tailMisses", C.c_uint64),
                ("tailAlphaTests", C.c_uint64), ("launchesTail", C.c_uint64), ("numMergedTriangles", C.c_uint64),
                ("msTraceFused", C.c_double), ("launchesTraceFused", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


assert C.sizeof(RtxState) == 48 and C.sizeof(SceneCamera) == 140 and C.sizeof(VertexAttributes) == 32
assert C.sizeof(GltfShadeMaterial) == 216 and C.sizeof(Light) == 64 and C.sizeof(EnvAccel) == 16
assert C.sizeof(Tonemapper) == 48 and C.sizeof(SunAndSky) == 96 and C.sizeof(PrimMesh) == 20 and C.sizeof(Node) == 68

vertex_dtype = np.dtype(VertexAttributes)
material_dtype = np.dtype(GltfShadeMaterial)
light_dtype = np.dtype(Light)
primmesh_dtype = np.dtype(PrimMesh)
node_dtype = np.dtype(Node)
envaccel_dtype = np.dtype(EnvAccel)


def default_rtx_state():
    """SampleExample::m_rtxState defaults (reference: src/sample_example.hpp:162-174)."""
    return RtxState(frame=0, maxDepth=10, maxSamples=1, fireflyClampThreshold=1.0, hdrMultiplier=1.0,
                    debugging_mode=0, pbrMode=0, _pad0=0, size=(0, 0), minHeatmap=0, maxHeatmap=65000)


def default_sun_and_sky():
    """SampleExample::m_sunAndSky defaults (reference: src/sample_example.hpp:176-193)."""
    return SunAndSky(rgb_unit_conversion=(1, 1, 1), multiplier=0.0000101320, haze=0.0, redblueshift=0.0, saturation=1.0,
                     horizon_height=0.0, ground_color=(0.4, 0.4, 0.4), horizon_blur=0.1, night_color=(0.0, 0.0, 0.01),
                     sun_disk_intensity=0.8, sun_direction=(0.00, 0.78, 0.62), sun_disk_scale=5.0,
                     sun_glow_intensity=1.0, y_is_up=1, physically_scaled_sun=1, in_use=0)


def default_tonemapper():
    """RenderOutput::m_tonemapper defaults (reference: src/render_output.hpp:37-49)."""
    return Tonemapper(brightness=1.0, contrast=1.0, saturation=1.0, vignette=0.0, avgLum=1.0, zoom=1.0,
                      renderingRatio=(1.0, 1.0), autoExposure=0, Ywhite=0.5, key=0.5, dither=1)


def default_material():
    """A glTF default material as nvh::GltfScene::importMaterials leaves it before
    Scene::createMaterialBuffer copies it (reference: src/scene.cpp:344-378)."""
    m = np.zeros((), dtype=material_dtype)
    m["pbrBaseColorFactor"] = (1, 1, 1, 1)
    m["pbrBaseColorTexture"] = -1
    m["pbrMetallicFactor"] = 1.0
    m["pbrRoughnessFactor"] = 1.0
    m["pbrMetallicRoughnessTexture"] = -1
    m["emissiveTexture"] = -1
    m["alphaMode"] = ALPHA_OPAQUE
    m["alphaCutoff"] = 0.5
    m["normalTexture"] = -1
    m["normalTextureScale"] = 1.0
    m["uvTransform"] = np.eye(4, dtype=np.float32).reshape(16)
    m["transmissionTexture"] = -1
    m["ior"] = 1.5
    m["anisotropyDirection"] = (0.0, 1.0, 0.0)  # (sin 0, cos 0, 0), src/scene.cpp:365
    m["attenuationColor"] = (1, 1, 1)
    m["thicknessTexture"] = -1
    m["attenuationDistance"] = np.float32(3.4028235e38)
    m["clearcoatTexture"] = -1
    m["clearcoatRoughnessTexture"] = -1
    return m
