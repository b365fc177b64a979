"""Mints the integer known-answer vectors in tests/golden/kat.npz.

The reference ships no test vectors, so these are produced by an INDEPENDENT numpy
implementation written directly from the reference GLSL / C++ (not from oracle/ and not from the HIP
code), and both the oracle and the product are checked against them:

  tea, pcg, rand                shaders/random.glsl:34-48, 59-65, 98-102
  pcg3d                         shaders/random.glsl:82-93
  compress_unit_vec (host)      shaders/compress.glsl:70-94, 111-139
  decompress_unit_vec           shaders/compress.glsl:142-180 (integer part + pre-normalisation floats)
  OffsetRay                     shaders/common.glsl:98-113
  handedness-bit packing        src/scene.cpp:232-239
  packUnorm4x8                  shaders/compress.glsl:58-72

Run:  python tests/golden/gen_kat.py   (rewrites kat.npz; deterministic)
"""
import os

import numpy as np

U32 = np.uint32


def tea(v0, v1):
    v0 = np.asarray(v0, dtype=np.uint64).copy()
    v1 = np.asarray(v1, dtype=np.uint64).copy()
    M = np.uint64(0xFFFFFFFF)
    s0 = np.uint64(0)
    for _ in range(16):
        s0 = (s0 + np.uint64(0x9E3779B9)) & M
        v0 = (v0 + ((((v1 << np.uint64(4)) & M) + np.uint64(0xA341316C) & M) ^ ((v1 + s0) & M) ^ (((v1 >> np.uint64(5)) + np.uint64(0xC8013EA4)) & M))) & M
        v1 = (v1 + ((((v0 << np.uint64(4)) & M) + np.uint64(0xAD90777D) & M) ^ ((v0 + s0) & M) ^ (((v0 >> np.uint64(5)) + np.uint64(0x7E95761E)) & M))) & M
    return v0.astype(U32)


def pcg_step(state):
    """returns (new_state, output word) for uint32 arrays"""
    s = np.asarray(state, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    prev = (s * np.uint64(747796405) + np.uint64(2891336453)) & M
    word = (((prev >> ((prev >> np.uint64(28)) + np.uint64(4))) ^ prev) * np.uint64(277803737)) & M
    return prev.astype(U32), (((word >> np.uint64(22)) ^ word) & M).astype(U32)


def word_to_float(w):
    bits = (np.uint32(0x3F800000) | (np.asarray(w, U32) >> np.uint32(9))).astype(U32)
    return bits.view(np.float32) - np.float32(1.0)


def pcg3d(v):
    v = np.asarray(v, dtype=np.uint64).copy()
    M = np.uint64(0xFFFFFFFF)
    v = (v * np.uint64(1664525) + np.uint64(1013904223)) & M
    v[:, 0] = (v[:, 0] + v[:, 1] * v[:, 2]) & M
    v[:, 1] = (v[:, 1] + v[:, 2] * v[:, 0]) & M
    v[:, 2] = (v[:, 2] + v[:, 0] * v[:, 1]) & M
    v ^= v >> np.uint64(16)
    v[:, 0] = (v[:, 0] + v[:, 1] * v[:, 2]) & M
    v[:, 1] = (v[:, 1] + v[:, 2] * v[:, 0]) & M
    v[:, 2] = (v[:, 2] + v[:, 0] * v[:, 1]) & M
    return v.astype(U32)


def round_half_even(x):
    # numpy's rint is IEEE round-half-to-even, which is what the reference's shim implements
    return np.rint(x.astype(np.float32)).astype(np.float32)


def compress_unit_vec(n):
    n = np.asarray(n, np.float32)
    d = np.float32(32767.0) / (np.abs(n[:, 0]) + np.abs(n[:, 1]) + np.abs(n[:, 2])).astype(np.float32)
    x = round_half_even(n[:, 0] * d).astype(np.int64)
    y = round_half_even(n[:, 1] * d).astype(np.int64)
    neg = n[:, 2] < 0
    mx = np.where(x < 0, -1, 0)
    my = np.where(y < 0, -1, 0)
    t = 32767 + mx + my
    fx = (t - (y ^ my)) ^ mx
    fy = (t - (x ^ mx)) ^ my
    x = np.where(neg, fx, x)
    y = np.where(neg, fy, y)
    packed = (((y + 32767) & 0xFFFF) << 16) | ((x + 32767) & 0xFFFF)
    packed = np.where(packed == 0xFFFFFFFF, 0xFFFFFFFE, packed)
    return packed.astype(U32)


def s2f(v):
    v = np.asarray(v, np.int64)
    pos = ((np.uint32(0x3F800000) | (np.abs(v).astype(U32) << np.uint32(8))).astype(U32)).view(np.float32)
    return np.where(v >= 0, pos - np.float32(1.0), -pos + np.float32(1.0)).astype(np.float32)


def decompress_prenorm(p):
    p = np.asarray(p, np.int64)
    x = (p & 0xFFFF) - 32767
    y = (p >> 16) - 32767
    mx = np.where(x < 0, -1, 0)
    my = np.where(y < 0, -1, 0)
    t0 = 32767 + mx + my
    ym = y ^ my
    t1 = t0 - (x ^ mx)
    z = t1 - ym
    x2 = np.where(z < 0, (t0 - ym) ^ mx, x)
    y2 = np.where(z < 0, t1 ^ my, y)
    return np.stack([s2f(x2), s2f(y2), s2f(z)], 1)


def offset_ray(p, n):
    p = np.asarray(p, np.float32)
    n = np.asarray(n, np.float32)
    of_i = (np.float32(256.0) * n).astype(np.int32)  # C truncation toward zero
    pi = (p.view(np.int32) + np.where(p < 0, -of_i, of_i)).astype(np.int32).view(np.float32)
    return np.where(np.abs(p) < np.float32(1.0 / 32.0), p + np.float32(1.0 / 65536.0) * n, pi).astype(np.float32)


def main():
    rng = np.random.default_rng(0x5EED0000)
    out = {}
    a = rng.integers(0, 2**32, 256, dtype=np.uint64).astype(U32)
    b = rng.integers(0, 2**32, 256, dtype=np.uint64).astype(U32)
    a[:4] = [0, 1, 0xFFFFFFFF, 1920 * 1079 + 1919]
    b[:4] = [0, 0, 0xFFFFFFFF, 255]
    out["tea_a"], out["tea_b"], out["tea_out"] = a, b, tea(a, b)

    seeds = tea(np.arange(8, dtype=np.uint64), np.full(8, 3, np.uint64))
    words = np.zeros((8, 32), U32)
    floats = np.zeros((8, 32), np.float32)
    st = seeds.copy()
    for i in range(32):
        st, w = pcg_step(st)
        words[:, i] = w
        floats[:, i] = word_to_float(w)
    out["pcg_seed"], out["pcg_words"], out["pcg_floats"], out["pcg_final"] = seeds, words, floats, st

    v = rng.integers(0, 2**32, (64, 3), dtype=np.uint64).astype(U32)
    v[0] = (0, 0, 0)
    v[1] = (1919, 1079, 0)
    out["pcg3d_in"], out["pcg3d_out"] = v, pcg3d(v)

    n = rng.normal(size=(512, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True).astype(np.float32)
    n[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    n[6] = [0.5, 0.5, -0.70710677]
    out["oct_in"] = n.astype(np.float32)
    out["oct_packed"] = compress_unit_vec(n)
    out["oct_prenorm"] = decompress_prenorm(out["oct_packed"])

    p = (rng.normal(size=(256, 3)) * np.array([10.0, 0.02, 300.0])).astype(np.float32)
    nn = rng.normal(size=(256, 3)).astype(np.float32)
    nn /= np.linalg.norm(nn, axis=1, keepdims=True).astype(np.float32)
    out["offs_p"], out["offs_n"], out["offs_out"] = p, nn, offset_ray(p, nn)

    uvy = rng.random(64).astype(np.float32) * 4 - 2
    hand = np.where(rng.random(64) < 0.5, -1.0, 1.0).astype(np.float32)
    bits = uvy.view(U32)
    out["hand_v"], out["hand_w"] = uvy, hand
    out["hand_out"] = np.where(hand > 0, bits | U32(1), bits & ~U32(1)).astype(U32)

    c = rng.random((64, 4)).astype(np.float32) * 1.4 - 0.2
    q = np.floor((np.clip(c, 0, 1).astype(np.float32) * np.float32(255.0)).astype(np.float64) + 0.5).astype(U32)  # std::round of a non-negative value
    out["unorm_in"] = c
    out["unorm_out"] = (q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16) | (q[:, 3] << 24)).astype(U32)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
