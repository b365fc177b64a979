"""Radiance .hdr environment loading (pt_hdr_load, vk_raytrace_amd/csrc/pt_hdr.cpp): replaces stbi_loadf in HdrSampling::loadEnvironment
(src/hdr_sampling.cpp:56-99).  Decoded texels against an independent numpy RGBE decode; and, where the reference tree is available, the
reference's OWN loadEnvironment (compiled unmodified, oracle/ref_glue/ref_host.cpp) run on top of this decoder: the texels, the sampler and
the EnvAccel buffer it uploads equal what the product computes for the same file."""
import ctypes as C
import os

import numpy as np
import pytest

from vk_raytrace_amd import capi, host_device as hd, synth


def to_rgbe(img):
    """float RGB -> RGBE bytes (the standard Radiance encoding)"""
    rgb = np.maximum(np.asarray(img, np.float64)[..., :3], 0.0)
    m = rgb.max(-1)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-300))) + 1, 0)
    scale = np.where(m > 1e-32, 256.0 / 2.0 ** e, 0.0)
    out = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    out[..., :3] = np.clip(np.floor(rgb * scale[..., None]), 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    return out


def rle_channel(row):
    out = bytearray()
    i, n = 0, len(row)
    while i < n:
        run = 1
        while i + run < n and run < 127 and row[i + run] == row[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, row[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            r = 1
            while j + r < n and r < 4 and row[j + r] == row[j]:
                r += 1
            if r >= 4:
                break
            j += 1
        out += bytes([j - i]) + bytes(row[i:j])
        i = j
    return bytes(out)


def write_hdr(path, rgbe, rle=True, magic=b"#?RADIANCE", extra=b"EXPOSURE=1.0\n"):
    h, w = rgbe.shape[:2]
    body = bytearray()
    for y in range(h):
        if rle and 8 <= w < 32768:
            body += bytes([2, 2, w >> 8, w & 255])
            for k in range(4):
                body += rle_channel(rgbe[y, :, k].tolist())
        else:
            body += rgbe[y].tobytes()
    with open(path, "wb") as f:
        f.write(magic + b"\n" + extra + b"FORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode() + bytes(body))


def numpy_decode(rgbe):
    f = np.where(rgbe[..., 3:4] != 0, np.ldexp(np.float32(1.0), rgbe[..., 3:4].astype(np.int32) - 136).astype(np.float32), np.float32(0))
    out = np.ones(rgbe.shape[:2] + (4,), np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * f
    return out


@pytest.mark.parametrize("shape,rle", [((16, 32), True), ((16, 32), False), ((9, 13), True), ((5, 7), True), ((3, 200), True), ((1, 8), True)])
def test_decode_matches_numpy(tmp_path, shape, rle):
    h, w = shape
    img = synth.procedural_sky(max(w, 8), max(h, 4))[:h, :w]
    img[0, 0, :3] = 0.0                       # a black texel (exponent 0)
    img[h // 2, : w // 2, :3] = 0.25           # a long run
    rgbe = to_rgbe(img)
    path = str(tmp_path / "env.hdr")
    write_hdr(path, rgbe, rle)
    got = capi.load_hdr(path)
    assert got.shape == (h, w, 4) and np.array_equal(got.view(np.uint32), numpy_decode(rgbe).view(np.uint32))
    assert (np.abs(got[..., :3] - img[..., :3]).max(-1) <= img[..., :3].max(-1) / 128 + 1e-6).all()   # 8-bit mantissa on the shared exponent


def test_rejects_malformed_files(tmp_path):
    rgbe = to_rgbe(synth.procedural_sky(16, 8))
    good = str(tmp_path / "good.hdr")
    write_hdr(good, rgbe, magic=b"#?RGBE")
    assert capi.load_hdr(good).shape == (8, 16, 4)
    data = open(good, "rb").read()
    cases = {"magic.hdr": data.replace(b"#?RGBE", b"#?NOPE!"), "format.hdr": data.replace(b"32-bit_rle_rgbe", b"32-bit_rle_xyze"), "layout.hdr": data.replace(b"-Y 8 +X 16", b"+X 16 -Y 8"),
             "short.hdr": data[:-20], "size.hdr": data.replace(b"-Y 8 +X 16", b"-Y 0 +X 16"), "run.hdr": data[:data.index(b"\x02\x02\x00\x10") + 4] + bytes([128 + 100, 7]) * 50}
    for name, blob in cases.items():
        p = tmp_path / name
        p.write_bytes(blob)
        with pytest.raises(capi.PtError):
            capi.load_hdr(str(p))
    with pytest.raises(capi.PtError):
        capi.load_hdr(str(tmp_path / "missing.hdr"))


def test_reference_load_environment_on_top_of_this_decoder(tmp_path):
    from tests import ref
    if not ref.available():
        pytest.skip("needs /root/reference (or a prebuilt oracle/_ref/libref.so)")
    R, P = ref.lib(), capi.lib()
    rgbe = to_rgbe(synth.procedural_sky(64, 32))
    path = str(tmp_path / "sky.hdr")
    write_hdr(path, rgbe)
    LOADER = C.CFUNCTYPE(C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int))

    @LOADER
    def loader(p, w, h):   # stands in for stbi_loadf; the pixels are malloc'ed by pt_hdr_load and released by the reference's stbi_image_free
        px = C.POINTER(C.c_float)()
        rc = P.pt_hdr_load(p, C.byref(px), w, h, None, 0)
        return C.cast(px, C.c_void_p).value if rc == 0 else None

    w, h = C.c_int(), C.c_int()
    texels = np.zeros((32, 64, 4), np.float32)
    accel = np.zeros(64 * 32, hd.envaccel_dtype)
    integral, average = C.c_float(), C.c_float()
    sampler = (C.c_int * 4)()
    R.ref_load_environment.argtypes = [C.c_char_p, LOADER, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]
    assert R.ref_load_environment(path.encode(), loader, texels.ctypes.data, C.byref(w), C.byref(h), accel.ctypes.data, C.byref(integral), C.byref(average), sampler) == 0
    assert (w.value, h.value) == (64, 32)
    mine = capi.load_hdr(path)
    assert np.array_equal(texels.view(np.uint32), mine.view(np.uint32))
    # the sampler the reference creates for the environment (LINEAR / LINEAR, U repeat, V clamp-to-edge) is the one pt_set_env assumes
    assert list(sampler) == [1, 1, 0, 2]
    acc, i2, a2 = capi.build_env_accel(mine)
    assert acc.tobytes() == accel.tobytes() and np.float32(i2) == np.float32(integral.value) and np.float32(a2) == np.float32(average.value)
