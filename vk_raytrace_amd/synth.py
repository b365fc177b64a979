"""Deterministic procedural stand-ins for the benchmark assets.

The reference downloads its assets at configure time (CMakeLists.txt:51-52) and none of the scenes
named in BASELINE.json (Sponza, DamagedHelmet, Bistro) nor any .hdr exist on the build or GPU box,
so every configuration is rendered on a seeded synthetic look-alike of the same triangle count,
material count and feature mix.  Everything here is closed-form / seeded numpy.
"""
import numpy as np

from . import host_device as hd
from .scene import Scene, Camera, translate, scale, rotate_x, rotate_y, rotate_z

SEED_BASE = 0x5EED0000


# ------------------------------------------------------------------------------------------------
# geometry primitives: return (pos, nrm, uv, idx, tan4)
# ------------------------------------------------------------------------------------------------
def grid(nx, ny, origin, du, dv, uv_scale=(1.0, 1.0), height=None):
    """(nx x ny) quads spanning origin + s*du + t*dv, normal = normalize(cross(du, dv)); optional
    height(s, t) displacement along the normal."""
    origin, du, dv = (np.asarray(a, np.float64) for a in (origin, du, dv))
    s, t = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), indexing="xy")
    n = np.cross(du, dv)
    n /= np.linalg.norm(n)
    pos = origin + s[..., None] * du + t[..., None] * dv
    nrm = np.broadcast_to(n, pos.shape).copy()
    if height is not None:
        h = height(s, t)
        pos = pos + h[..., None] * n
        # finite-difference normals
        eps = 1e-3
        hs = (height(s + eps, t) - h) / eps
        ht = (height(s, t + eps) - h) / eps
        tu = du + hs[..., None] * n
        tv = dv + ht[..., None] * n
        nrm = np.cross(tu, tv)
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    uv = np.stack([s * uv_scale[0], t * uv_scale[1]], -1)
    tdir = du / np.linalg.norm(du)
    tan = np.concatenate([np.broadcast_to(tdir, pos.shape), np.ones(pos.shape[:-1] + (1,))], -1)
    i = (np.arange(ny)[:, None] * (nx + 1) + np.arange(nx)[None, :]).reshape(-1)
    idx = np.stack([i, i + 1, i + nx + 2, i, i + nx + 2, i + nx + 1], -1).reshape(-1)
    return (pos.reshape(-1, 3).astype(np.float32), nrm.reshape(-1, 3).astype(np.float32), uv.reshape(-1, 2).astype(np.float32),
            idx.astype(np.uint32), tan.reshape(-1, 4).astype(np.float32))


def merge(parts):
    pos, nrm, uv, idx, tan = [], [], [], [], []
    base = 0
    for p, n, u, i, t in parts:
        pos.append(p); nrm.append(n); uv.append(u); tan.append(t); idx.append(i + base)
        base += len(p)
    return np.concatenate(pos), np.concatenate(nrm), np.concatenate(uv), np.concatenate(idx).astype(np.uint32), np.concatenate(tan)


def box(size=(1, 1, 1), sub=1, inward=False):
    sx, sy, sz = (0.5 * s for s in size)
    faces = [((-sx, -sy, sz), (2 * sx, 0, 0), (0, 2 * sy, 0)),   # +z
             ((sx, -sy, -sz), (-2 * sx, 0, 0), (0, 2 * sy, 0)),  # -z
             ((sx, -sy, sz), (0, 0, -2 * sz), (0, 2 * sy, 0)),   # +x
             ((-sx, -sy, -sz), (0, 0, 2 * sz), (0, 2 * sy, 0)),  # -x
             ((-sx, sy, sz), (2 * sx, 0, 0), (0, 0, -2 * sz)),   # +y
             ((-sx, -sy, -sz), (2 * sx, 0, 0), (0, 0, 2 * sz))]  # -y
    parts = []
    for o, du, dv in faces:
        if inward:
            parts.append(grid(sub, sub, np.add(o, du), np.negative(du), dv))
        else:
            parts.append(grid(sub, sub, o, du, dv))
    return merge(parts)


def revolve(profile_r, profile_y, segments, v_scale=1.0, u_scale=1.0):
    """Surface of revolution about +Y from a (radius, y) profile going bottom -> top."""
    r = np.asarray(profile_r, np.float64)
    y = np.asarray(profile_y, np.float64)
    ang = np.linspace(0, 2 * np.pi, segments + 1)
    ca, sa = np.cos(ang), np.sin(ang)
    pos = np.stack([r[:, None] * ca[None, :], np.broadcast_to(y[:, None], (len(r), segments + 1)), r[:, None] * sa[None, :]], -1)
    dr = np.gradient(r)
    dy = np.gradient(y)
    ln = np.maximum(np.hypot(dr, dy), 1e-12)
    nr, ny_ = dy / ln, -dr / ln
    nrm = np.stack([nr[:, None] * ca[None, :], np.broadcast_to(ny_[:, None], (len(r), segments + 1)), nr[:, None] * sa[None, :]], -1)
    tan = np.stack([-sa, np.zeros_like(sa), ca, np.ones_like(sa)], -1)
    tan = np.broadcast_to(tan[None], (len(r), segments + 1, 4))
    uu = np.broadcast_to((ang / (2 * np.pi) * u_scale)[None, :], (len(r), segments + 1))
    vv = np.broadcast_to((np.linspace(0, 1, len(r)) * v_scale)[:, None], (len(r), segments + 1))
    uv = np.stack([uu, vv], -1)
    rows, cols = len(r) - 1, segments
    i = (np.arange(rows)[:, None] * (segments + 1) + np.arange(cols)[None, :]).reshape(-1)
    # winding chosen so that cross(p1-p0, p2-p0) points away from the axis
    idx = np.stack([i, i + segments + 1, i + segments + 2, i, i + segments + 2, i + 1], -1).reshape(-1)
    return (pos.reshape(-1, 3).astype(np.float32), nrm.reshape(-1, 3).astype(np.float32), uv.reshape(-1, 2).astype(np.float32),
            idx.astype(np.uint32), tan.reshape(-1, 4).astype(np.float32))


def uv_sphere(radius, nu, nv, displace=None):
    th = np.linspace(0, np.pi, nv + 1)
    ph = np.linspace(0, 2 * np.pi, nu + 1)
    st, ct = np.sin(th)[:, None], np.cos(th)[:, None]
    cp, sp = np.cos(ph)[None, :], np.sin(ph)[None, :]
    d = np.stack([st * cp, np.broadcast_to(ct, (nv + 1, nu + 1)), st * sp], -1)
    rad = np.full(d.shape[:2], radius, np.float64)
    if displace is not None:
        rad = rad + displace(d)
    pos = d * rad[..., None]
    nrm = d.copy()
    if displace is not None:
        # normals from the displaced surface by central differences of the position field
        pu = np.gradient(pos, axis=1)
        pv = np.gradient(pos, axis=0)
        nn = np.cross(pu, pv)
        ln = np.linalg.norm(nn, axis=-1, keepdims=True)
        good = ln[..., 0] > 1e-12
        nn[good] /= ln[good]
        nn[~good] = d[~good]
        flip = np.sum(nn * d, -1) < 0
        nn[flip] *= -1
        nrm = nn
    tan = np.stack([np.broadcast_to(-sp, (nv + 1, nu + 1)), np.zeros((nv + 1, nu + 1)), np.broadcast_to(cp, (nv + 1, nu + 1)), np.ones((nv + 1, nu + 1))], -1)
    uv = np.stack([np.broadcast_to(ph / (2 * np.pi), (nv + 1, nu + 1)), np.broadcast_to((th / np.pi)[:, None], (nv + 1, nu + 1))], -1)
    i = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    idx = np.stack([i, i + 1, i + nu + 2, i, i + nu + 2, i + nu + 1], -1).reshape(-1)
    return (pos.reshape(-1, 3).astype(np.float32), nrm.reshape(-1, 3).astype(np.float32), uv.reshape(-1, 2).astype(np.float32),
            idx.astype(np.uint32), tan.reshape(-1, 4).astype(np.float32))


def cards(rng, count, center, spread, size, up_bias=0.6):
    """`count` randomly oriented quads (foliage cards); uv covers 0..1 per card."""
    c = np.asarray(center, np.float64) + (rng.random((count, 3)) - 0.5) * np.asarray(spread)
    n = rng.normal(size=(count, 3))
    n[:, 1] = np.abs(n[:, 1]) * up_bias
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    ref = np.where(np.abs(n[:, 1:2]) < 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
    t = np.cross(ref, n)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    s = size * (0.6 + 0.8 * rng.random((count, 1)))
    corners = np.stack([c - t * s - b * s, c + t * s - b * s, c + t * s + b * s, c - t * s + b * s], 1)  # (count,4,3)
    pos = corners.reshape(-1, 3)
    nrm = np.repeat(n, 4, axis=0)
    tan = np.concatenate([np.repeat(t, 4, axis=0), np.ones((count * 4, 1))], 1)
    uv = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float64), (count, 1))
    base = (np.arange(count) * 4)[:, None]
    idx = (base + np.array([[0, 1, 2, 0, 2, 3]])).reshape(-1)
    return pos.astype(np.float32), nrm.astype(np.float32), uv.astype(np.float32), idx.astype(np.uint32), tan.astype(np.float32)


def transform_mesh(mesh, m):
    pos, nrm, uv, idx, tan = mesh
    m = np.asarray(m, np.float64)
    p = pos @ m[:3, :3].T + m[:3, 3]
    nm = np.linalg.inv(m[:3, :3]).T
    n = nrm @ nm.T
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    t3 = tan[:, :3] @ m[:3, :3].T
    t3 /= np.maximum(np.linalg.norm(t3, axis=1, keepdims=True), 1e-20)
    t = np.concatenate([t3, tan[:, 3:4]], 1)
    if np.linalg.det(m[:3, :3]) < 0:
        idx = idx.reshape(-1, 3)[:, ::-1].reshape(-1)
    return p.astype(np.float32), n.astype(np.float32), uv, idx.astype(np.uint32), t.astype(np.float32)


# ------------------------------------------------------------------------------------------------
# textures
# ------------------------------------------------------------------------------------------------
def _fbm(rng, size, octaves=5, base=4):
    out = np.zeros((size, size), np.float64)
    amp, tot = 1.0, 0.0
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    for o in range(octaves):
        n = base * (2 ** o)
        g = rng.random((n, n))
        fx, fy = xx * n / size, yy * n / size
        x0, y0 = np.floor(fx).astype(int) % n, np.floor(fy).astype(int) % n
        x1, y1 = (x0 + 1) % n, (y0 + 1) % n
        ax, ay = fx - np.floor(fx), fy - np.floor(fy)
        ax, ay = ax * ax * (3 - 2 * ax), ay * ay * (3 - 2 * ay)
        v = (g[y0, x0] * (1 - ax) + g[y0, x1] * ax) * (1 - ay) + (g[y1, x0] * (1 - ax) + g[y1, x1] * ax) * ay
        out += amp * v
        tot += amp
        amp *= 0.5
    return out / tot


def _to_rgba8(rgb, a=None):
    rgb = np.clip(rgb, 0, 1)
    if a is None:
        a = np.ones(rgb.shape[:2])
    return (np.concatenate([rgb, np.clip(a, 0, 1)[..., None]], -1) * 255.0 + 0.5).astype(np.uint8)


def tex_albedo(rng, size, tint, contrast=0.5, pattern="noise"):
    n = _fbm(rng, size)
    if pattern == "brick":
        yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
        rows = (yy * 8 // size)
        bx = ((xx + (rows % 2) * size // 16) * 4 // size)
        mortar = (((yy * 8) % size) < size // 10) | ((((xx + (rows % 2) * size // 16) * 4) % size) < size // 16)
        n = np.where(mortar, 0.75, 0.35 + 0.4 * n + 0.1 * ((bx + rows) % 3))
    elif pattern == "checker":
        yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
        n = np.where(((xx * 8 // size) + (yy * 8 // size)) % 2 == 0, 0.85, 0.25) + 0.1 * (n - 0.5)
    v = (1 - contrast) + contrast * n
    return _to_rgba8(v[..., None] * np.asarray(tint)[None, None, :])


def tex_metal_rough(rng, size, rough=(0.3, 0.9), metal=(0.0, 1.0)):
    r = _fbm(rng, size)
    m = _fbm(rng, size, octaves=3)
    rgb = np.stack([np.ones_like(r), rough[0] + (rough[1] - rough[0]) * r, metal[0] + (metal[1] - metal[0]) * (m > 0.5)], -1)
    return _to_rgba8(rgb)


def tex_normal(rng, size, strength=2.0):
    h = _fbm(rng, size, octaves=6)
    dx = (np.roll(h, -1, 1) - np.roll(h, 1, 1)) * size / 64.0 * strength
    dy = (np.roll(h, -1, 0) - np.roll(h, 1, 0)) * size / 64.0 * strength
    n = np.stack([-dx, -dy, np.ones_like(h)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return _to_rgba8(n * 0.5 + 0.5)


def tex_emissive(rng, size, color=(1.0, 0.6, 0.2)):
    n = _fbm(rng, size, octaves=3, base=8)
    mask = (n > 0.62).astype(np.float64)
    return _to_rgba8(mask[..., None] * np.asarray(color)[None, None, :])


def tex_leaf(rng, size, tint=(0.25, 0.55, 0.15)):
    """Leaf-shaped alpha mask for ALPHA_MASK foliage cards."""
    yy, xx = np.meshgrid(np.linspace(-1, 1, size), np.linspace(-1, 1, size), indexing="ij")
    lobes = np.zeros_like(xx, bool)
    for k in range(5):
        a = k * 2 * np.pi / 5 + 0.3
        cx, cy = 0.45 * np.cos(a), 0.45 * np.sin(a)
        ca, sa = np.cos(a), np.sin(a)
        u = (xx - cx) * ca + (yy - cy) * sa
        v = -(xx - cx) * sa + (yy - cy) * ca
        lobes |= (u / 0.5) ** 2 + (v / 0.2) ** 2 < 1
    n = _fbm(rng, size, octaves=4)
    rgb = (0.6 + 0.4 * n)[..., None] * np.asarray(tint)[None, None, :]
    return _to_rgba8(rgb, lobes.astype(np.float64))


# ------------------------------------------------------------------------------------------------
# environment
# ------------------------------------------------------------------------------------------------
def constant_env(w=16, h=8, value=1.0):
    e = np.full((h, w, 4), value, np.float32)
    e[..., 3] = 1.0
    return e


def procedural_sky(w=2048, h=1024, sun_elevation_deg=45.0, sun_azimuth_deg=30.0, sun_peak=5e4, sun_sigma_deg=1.5):
    """Lat-long RGBA32F: vertical blue-ish gradient (0.3 -> 1.0) + Gaussian sun lobe.
    Row 0 is the +Y pole; u = atan2(z, x)/2pi + 0.5 (shaders/common.glsl:67-74)."""
    v = (np.arange(h) + 0.5) / h
    u = (np.arange(w) + 0.5) / w
    theta = v * np.pi  # 0 at +Y
    phi = u * 2 * np.pi - np.pi
    st, ct = np.sin(theta)[:, None], np.cos(theta)[:, None]
    d = np.stack([st * np.cos(phi)[None, :], np.broadcast_to(ct, (h, w)), st * np.sin(phi)[None, :]], -1)
    up = np.clip(d[..., 1] * 0.5 + 0.5, 0, 1)
    grad = 0.3 + 0.7 * up
    sky = grad[..., None] * np.array([0.55, 0.75, 1.0])[None, None, :]
    ground = np.array([0.25, 0.22, 0.2])[None, None, :] * (0.4 + 0.6 * up[..., None])
    col = np.where((d[..., 1] >= 0)[..., None], sky, ground)
    el, az = np.radians(sun_elevation_deg), np.radians(sun_azimuth_deg)
    sd = np.array([np.cos(el) * np.cos(az), np.sin(el), np.cos(el) * np.sin(az)])
    ang = np.arccos(np.clip(d @ sd, -1, 1))
    lobe = sun_peak * np.exp(-0.5 * (ang / np.radians(sun_sigma_deg)) ** 2)
    col = col + lobe[..., None] * np.array([1.0, 0.95, 0.85])[None, None, :]
    out = np.ones((h, w, 4), np.float32)
    out[..., :3] = col.astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------
# scenes
# ------------------------------------------------------------------------------------------------
def quad_scene():
    """C1: one double-sided quad, 2 triangles."""
    sc = Scene("quad")
    m = sc.add_material(pbrBaseColorFactor=(0.8, 0.8, 0.8, 1.0), pbrRoughnessFactor=1.0, pbrMetallicFactor=0.0, doubleSided=1)
    pos = [(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)]
    nrm = [(0, 0, 1)] * 4
    uv = [(0, 0), (1, 0), (1, 1), (0, 1)]
    pm = sc.add_prim_mesh(pos, nrm, uv, [0, 1, 2, 0, 2, 3], m, tangents=[(1, 0, 0, 1)] * 4)
    sc.add_node(pm)
    sc.camera = Camera(eye=(0, 0, 3), center=(0, 0, 0), up=(0, 1, 0), fov=45.0)
    return sc


def _add(sc, mesh, material, matrix=None, colors=None):
    pos, nrm, uv, idx, tan = mesh
    pm = sc.add_prim_mesh(pos, nrm, uv, idx, material, tangents=tan, colors=colors)
    sc.add_node(pm, matrix)
    return pm


def feature_box(tex_size=64, lights=False, seed=SEED_BASE + 100):
    """A small closed room exercising every material / trace feature the hot path has:
    textured + normal-mapped diffuse, metal, clearcoat, anisotropy, sheen, emissive (textured),
    thick + thin-walled transmission with volume absorption, unlit, ALPHA_MASK and ALPHA_BLEND cards,
    vertex colours, uvTransform, NEAREST / mirrored / clamped samplers, an instanced prim-mesh under a
    mirrored (negative-determinant) and a non-uniformly scaled transform, single- and double-sided
    geometry, optional punctual lights.  The top is open to the environment."""
    rng = np.random.default_rng(seed)
    sc = Scene("feature_box")
    t_wall = sc.add_texture(tex_albedo(rng, tex_size, (0.8, 0.75, 0.7), pattern="brick"))
    t_wall_n = sc.add_texture(tex_normal(rng, tex_size))
    t_floor = sc.add_texture(tex_albedo(rng, tex_size, (0.9, 0.9, 0.9), pattern="checker"), magFilter=hd.FILTER_NEAREST)
    t_mr = sc.add_texture(tex_metal_rough(rng, tex_size))
    t_em = sc.add_texture(tex_emissive(rng, tex_size), wrapS=hd.WRAP_MIRRORED_REPEAT, wrapT=hd.WRAP_CLAMP_TO_EDGE)
    t_leaf = sc.add_texture(tex_leaf(rng, tex_size))
    t_noise = sc.add_texture(tex_albedo(rng, tex_size, (1, 1, 1), contrast=1.0))

    uvt = np.eye(4, dtype=np.float32)
    uvt[0, 0], uvt[1, 1] = 2.0, 2.0
    uvt[2, 0], uvt[2, 1] = 0.25, 0.5  # nvh builds KHR_texture_transform in row-vector convention: the offset multiplies the z = 1 of (u, v, 1, 1) (Appendix C-16)

    m_wall = sc.add_material(pbrBaseColorTexture=t_wall, normalTexture=t_wall_n, pbrMetallicFactor=0.0, pbrRoughnessFactor=0.9, normalTextureScale=1.0)
    m_floor = sc.add_material(pbrBaseColorTexture=t_floor, pbrMetallicFactor=0.0, pbrRoughnessFactor=0.6, uvTransform=uvt.T.reshape(16))
    m_red = sc.add_material(pbrBaseColorFactor=(0.75, 0.15, 0.1, 1), pbrMetallicFactor=0.0, pbrRoughnessFactor=1.0)
    m_green = sc.add_material(pbrBaseColorFactor=(0.15, 0.7, 0.2, 1), pbrMetallicFactor=0.0, pbrRoughnessFactor=1.0, doubleSided=1)
    m_metal = sc.add_material(pbrBaseColorFactor=(0.95, 0.8, 0.4, 1), pbrMetallicRoughnessTexture=t_mr, pbrMetallicFactor=1.0, pbrRoughnessFactor=0.6)
    m_coat = sc.add_material(pbrBaseColorFactor=(0.1, 0.2, 0.8, 1), pbrMetallicFactor=0.0, pbrRoughnessFactor=0.5, clearcoatFactor=1.0,
                             clearcoatRoughness=0.1, clearcoatTexture=t_noise, clearcoatRoughnessTexture=t_mr)
    m_aniso = sc.add_material(pbrBaseColorFactor=(0.9, 0.9, 0.9, 1), pbrMetallicFactor=1.0, pbrRoughnessFactor=0.35, anisotropy=0.8,
                              anisotropyDirection=(np.sin(0.6), np.cos(0.6), 0.0))
    m_glass = sc.add_material(pbrBaseColorFactor=(1, 1, 1, 1), pbrMetallicFactor=0.0, pbrRoughnessFactor=0.05, transmissionFactor=1.0, ior=1.5,
                              thicknessFactor=1.0, attenuationColor=(0.6, 0.9, 0.7), attenuationDistance=0.5, doubleSided=1)
    m_thin = sc.add_material(pbrBaseColorFactor=(0.9, 0.95, 1.0, 1), pbrMetallicFactor=0.0, pbrRoughnessFactor=0.2, transmissionFactor=0.8,
                             transmissionTexture=t_noise, thicknessFactor=0.0, doubleSided=1)
    m_emit = sc.add_material(pbrBaseColorFactor=(0.1, 0.1, 0.1, 1), emissiveFactor=(6.0, 5.0, 4.0), emissiveTexture=t_em, pbrMetallicFactor=0.0)
    m_unlit = sc.add_material(pbrBaseColorFactor=(0.2, 0.9, 0.9, 1), unlit=1)
    m_leaf = sc.add_material(pbrBaseColorTexture=t_leaf, alphaMode=hd.ALPHA_MASK, alphaCutoff=0.5, doubleSided=1, pbrMetallicFactor=0.0, pbrRoughnessFactor=0.8)
    m_blend = sc.add_material(pbrBaseColorFactor=(0.9, 0.3, 0.8, 0.45), alphaMode=hd.ALPHA_BLEND, doubleSided=1, pbrMetallicFactor=0.0, pbrRoughnessFactor=0.7)
    sheen_bits = int(0.8 * 255 + 0.5) | (int(0.3 * 255 + 0.5) << 8) | (int(0.9 * 255 + 0.5) << 16) | (int(0.5 * 255 + 0.5) << 24)
    m_sheen = sc.add_material(pbrBaseColorFactor=(0.4, 0.05, 0.3, 1), pbrMetallicFactor=0.0, pbrRoughnessFactor=0.9, sheen=sheen_bits)

    # room 4 x 3 x 4, open top, single-sided walls facing inward
    _add(sc, grid(8, 8, (-2, 0, 2), (4, 0, 0), (0, 0, -4), uv_scale=(2, 2)), m_floor)
    _add(sc, grid(8, 6, (-2, 0, -2), (4, 0, 0), (0, 3, 0), uv_scale=(2, 1.5)), m_wall)       # back wall, normal +z
    _add(sc, grid(8, 6, (-2, 0, 2), (0, 0, -4), (0, 3, 0)), m_red)                            # left wall, normal +x
    _add(sc, grid(8, 6, (2, 0, -2), (0, 0, 4), (0, 3, 0)), m_green)                           # right wall, normal -x
    # objects
    sph = uv_sphere(0.45, 32, 16)
    pm_sph = sc.add_prim_mesh(*sph[:4], m_metal, tangents=sph[4])
    sc.add_node(pm_sph, translate(-1.2, 0.45, -0.8))
    # the same prim-mesh instanced under a mirrored transform (negative determinant) and a stretch
    sc.add_node(pm_sph, translate(1.3, 0.5, -1.2) @ scale(-1.0, 1.1, 0.8))
    _add(sc, uv_sphere(0.4, 32, 16), m_coat, translate(0.0, 0.4, -1.3))
    _add(sc, uv_sphere(0.35, 32, 16), m_glass, translate(-0.5, 0.35, 0.4))
    _add(sc, uv_sphere(0.3, 24, 12), m_sheen, translate(1.2, 0.3, 0.5))
    prof_y = np.linspace(0, 1, 24)
    prof_r = 0.18 + 0.1 * np.sin(prof_y * 5.0)
    _add(sc, revolve(prof_r, prof_y, 32), m_aniso, translate(0.6, 0.0, 0.2))
    _add(sc, box((0.5, 0.5, 0.5), sub=2), m_unlit, translate(-1.5, 0.25, 0.9) @ rotate_y(0.5))
    _add(sc, grid(2, 2, (-0.4, 2.2, -1.6), (0.8, 0, 0), (0, 0, 0.8)), m_emit)  # emissive panel facing down (normal -y): du x dv = +x x +z = -y
    _add(sc, grid(1, 1, (-0.2, 0.1, 1.0), (0.9, 0, 0), (0, 0.9, 0)), m_thin)   # thin pane
    # vertex-coloured box
    b = box((0.4, 0.8, 0.4), sub=1)
    col = np.ones((len(b[0]), 4), np.float32)
    col[:, :3] = 0.5 + 0.5 * np.sign(b[0]) * np.array([1.0, 0.5, 0.25])
    _add(sc, b, m_red, translate(1.5, 0.4, 1.2), colors=np.clip(col, 0, 1))
    # alpha-tested foliage and blended panes between camera and objects
    _add(sc, cards(rng, 60, (-0.2, 0.9, 0.9), (2.4, 1.2, 0.8), 0.22), m_leaf)
    _add(sc, grid(1, 1, (0.3, 0.2, 1.3), (0.8, 0, 0.2), (0, 1.0, 0)), m_blend)
    _add(sc, grid(1, 1, (0.1, 0.3, 1.5), (0.8, 0, -0.1), (0, 0.8, 0)), m_blend)
    if lights:
        sc.add_light(type=hd.LightType_Point, position=(0.0, 2.5, 0.5), color=(1.0, 0.9, 0.8), intensity=8.0, range=0.0)
        sc.add_light(type=hd.LightType_Spot, position=(-1.5, 2.6, 1.5), direction=(0.4, -1.0, -0.4), color=(0.6, 0.7, 1.0), intensity=20.0,
                     range=12.0, innerConeCos=np.cos(0.3), outerConeCos=np.cos(0.5))
        sc.add_light(type=hd.LightType_Directional, direction=(-0.3, -1.0, -0.2), color=(1, 1, 1), intensity=1.5)
    sc.camera = Camera(eye=(0.0, 1.4, 4.6), center=(0.0, 0.9, 0.0), up=(0, 1, 0), fov=50.0)
    return sc


def helmet_like(target_tris=70_000, tex_size=2048, seed=SEED_BASE + 2):
    """C2 stand-in for DamagedHelmet: one displaced UV-sphere, one material, 5 textures."""
    rng = np.random.default_rng(seed)
    sc = Scene("helmet_like")
    nv = int(np.sqrt(target_tris / 4.0))
    nu = 2 * nv
    k = rng.normal(size=(6, 3))

    def disp(d):
        h = np.zeros(d.shape[:2])
        for i in range(6):
            h += 0.03 / (1 + i) * np.sin((3 + 2 * i) * (d @ k[i]) + i)
        visor = np.exp(-((d[..., 2] - 0.8) ** 2 + (d[..., 1] - 0.1) ** 2) / 0.08)
        return h - 0.08 * visor

    t_alb = sc.add_texture(tex_albedo(rng, tex_size, (0.75, 0.7, 0.65), contrast=0.6))
    t_mr = sc.add_texture(tex_metal_rough(rng, tex_size, rough=(0.2, 0.8)))
    t_n = sc.add_texture(tex_normal(rng, tex_size, strength=1.5))
    t_e = sc.add_texture(tex_emissive(rng, tex_size, (0.4, 0.8, 1.0)))
    sc.add_texture(tex_albedo(rng, tex_size, (1, 1, 1)))  # the (unused) AO map of the asset
    m = sc.add_material(pbrBaseColorTexture=t_alb, pbrMetallicRoughnessTexture=t_mr, normalTexture=t_n, emissiveTexture=t_e,
                        emissiveFactor=(1.0, 1.0, 1.0), pbrMetallicFactor=1.0, pbrRoughnessFactor=1.0)
    _add(sc, uv_sphere(1.0, nu, nv, displace=disp), m)
    sc.camera = Camera(eye=(1.6, 0.9, 2.6), center=(0, 0, 0), up=(0, 1, 0), fov=45.0)
    return sc


def sponza_like(target_tris=262_267, tex_size=1024, seed=SEED_BASE + 3, foliage_frac=0.10):
    """C3/C4 stand-in for Crytek Sponza: colonnaded two-storey atrium open to the sky, ~25 materials,
    one albedo (+ optional normal / metal-rough) texture each, ~10 % of the triangles under ALPHA_MASK
    foliage cards so the any-hit path is exercised; camera inside looking down the nave."""
    rng = np.random.default_rng(seed)
    sc = Scene("sponza_like")
    L, Wd, Hh = 24.0, 10.0, 9.0  # nave length (x), width (z), height (y)

    def mat_textured(tint, pattern="noise", normal=True, mr=False, **kw):
        args = dict(pbrBaseColorTexture=sc.add_texture(tex_albedo(rng, tex_size, tint, pattern=pattern)), pbrMetallicFactor=0.0, pbrRoughnessFactor=0.85)
        if normal:
            args["normalTexture"] = sc.add_texture(tex_normal(rng, tex_size))
        if mr:
            args["pbrMetallicRoughnessTexture"] = sc.add_texture(tex_metal_rough(rng, tex_size, metal=(0.0, 0.0)))
            args["pbrRoughnessFactor"] = 1.0
        args.update(kw)
        return sc.add_material(**args)

    m_floor = mat_textured((0.75, 0.7, 0.62), "checker", mr=True)
    m_wall = [mat_textured(t, "brick", mr=(i % 2 == 0)) for i, t in enumerate([(0.8, 0.74, 0.66), (0.72, 0.68, 0.6), (0.78, 0.7, 0.58), (0.7, 0.66, 0.62)])]
    m_col = [mat_textured(t, "noise", mr=True) for t in [(0.82, 0.8, 0.75), (0.76, 0.74, 0.7), (0.7, 0.68, 0.66)]]
    m_arch = [mat_textured(t, "noise") for t in [(0.8, 0.76, 0.7), (0.74, 0.7, 0.64)]]
    m_cloth = [mat_textured(t, "noise", normal=False, doubleSided=1) for t in [(0.7, 0.1, 0.1), (0.1, 0.5, 0.15), (0.1, 0.2, 0.6)]]
    m_vase = [mat_textured(t, "noise", mr=True) for t in [(0.5, 0.35, 0.2), (0.3, 0.3, 0.35)]]
    m_metal = [sc.add_material(pbrBaseColorFactor=c + (1.0,), pbrMetallicFactor=1.0, pbrRoughnessFactor=r) for c, r in [((0.9, 0.7, 0.3), 0.35), ((0.8, 0.8, 0.85), 0.2)]]
    m_roof = mat_textured((0.6, 0.45, 0.35), "brick")
    leaf_tex = [sc.add_texture(tex_leaf(rng, tex_size, tint)) for tint in [(0.25, 0.55, 0.15), (0.35, 0.5, 0.1), (0.2, 0.45, 0.2)]]
    m_leaf = [sc.add_material(pbrBaseColorTexture=t, alphaMode=hd.ALPHA_MASK, alphaCutoff=0.5, doubleSided=1, pbrMetallicFactor=0.0, pbrRoughnessFactor=0.7) for t in leaf_tex]
    m_chain = sc.add_material(pbrBaseColorTexture=leaf_tex[0], pbrBaseColorFactor=(0.6, 0.6, 0.65, 1.0), alphaMode=hd.ALPHA_MASK, alphaCutoff=0.4,
                              doubleSided=1, pbrMetallicFactor=1.0, pbrRoughnessFactor=0.4)
    m_lion = mat_textured((0.7, 0.65, 0.55), "noise", mr=True)

    budget = {"tris": 0}

    def add(mesh, material, matrix=None):
        budget["tris"] += len(mesh[3]) // 3
        return _add(sc, mesh, material, matrix)

    # shell: floor, two long walls (two storeys each), two end walls, partial roof with central opening
    add(grid(96, 40, (-L / 2, 0, Wd / 2), (L, 0, 0), (0, 0, -Wd), uv_scale=(12, 5)), m_floor)
    for side, z in ((0, -Wd / 2), (1, Wd / 2)):
        for storey in range(2):
            y0 = storey * Hh / 2
            if side == 0:
                add(grid(96, 20, (-L / 2, y0, z), (L, 0, 0), (0, Hh / 2, 0), uv_scale=(12, 2.5)), m_wall[storey])
            else:
                add(grid(96, 20, (L / 2, y0, z), (-L, 0, 0), (0, Hh / 2, 0), uv_scale=(12, 2.5)), m_wall[2 + storey])
    add(grid(40, 36, (-L / 2, 0, Wd / 2), (0, 0, -Wd), (0, Hh, 0), uv_scale=(5, 4.5)), m_wall[0])
    add(grid(40, 36, (L / 2, 0, -Wd / 2), (0, 0, Wd), (0, Hh, 0), uv_scale=(5, 4.5)), m_wall[1])
    add(grid(96, 12, (-L / 2, Hh, -Wd / 2), (L, 0, 0), (0, 0, Wd * 0.3), uv_scale=(12, 1.5)), m_roof)   # roof strips, normal -y
    add(grid(96, 12, (-L / 2, Hh, Wd * 0.2), (L, 0, 0), (0, 0, Wd * 0.3), uv_scale=(12, 1.5)), m_roof)

    # colonnade: 2 rows x 10 columns x 2 storeys, fluted shafts + capitals + arches between them
    ncol = 10
    shaft_y = np.linspace(0, 1, 25)
    xs = np.linspace(-L / 2 + 1.5, L / 2 - 1.5, ncol)
    col_mesh = None
    for storey in range(2):
        for zi, z in enumerate((-Wd / 2 + 1.6, Wd / 2 - 1.6)):
            for ci, x in enumerate(xs):
                h = Hh / 2 - 0.9
                r = (0.34 - 0.06 * shaft_y) * (1 + 0.0 * shaft_y)
                shaft = revolve(r, shaft_y * h, 40, v_scale=3.0, u_scale=2.0)
                # fluting: modulate radius around the axis
                p = shaft[0].copy()
                ang = np.arctan2(p[:, 2], p[:, 0])
                f = 1.0 + 0.04 * np.cos(ang * 16)
                p[:, 0] *= f
                p[:, 2] *= f
                shaft = (p,) + shaft[1:]
                add(shaft, m_col[(ci + zi + storey) % 3], translate(x, storey * Hh / 2, z))
                add(box((0.9, 0.25, 0.9), sub=3), m_col[(ci + 1) % 3], translate(x, storey * Hh / 2 + h + 0.125, z))
                add(box((0.8, 0.2, 0.8), sub=2), m_col[(ci + 2) % 3], translate(x, storey * Hh / 2 + 0.1, z))
            # arches: half-rings between neighbouring columns
            for ci in range(ncol - 1):
                xm = 0.5 * (xs[ci] + xs[ci + 1])
                rad = 0.5 * (xs[ci + 1] - xs[ci]) - 0.35
                a = np.linspace(0, np.pi, 25)
                # sweep a small square cross-section along the half circle
                prof = np.array([[-0.15, -0.3], [0.15, -0.3], [0.15, 0.3], [-0.15, 0.3], [-0.15, -0.3]])
                pos = []
                for pr in prof:
                    rr = rad + pr[0]
                    pos.append(np.stack([xm + rr * np.cos(a), storey * Hh / 2 + (Hh / 2 - 0.9) + 0.25 + rr * np.sin(a), np.full_like(a, z + pr[1])], -1))
                pos = np.stack(pos, 0)  # (5, 25, 3)
                cen = np.stack([xm + rad * np.cos(a), storey * Hh / 2 + (Hh / 2 - 0.9) + 0.25 + rad * np.sin(a), np.full_like(a, z)], -1)
                nrm = pos - cen[None]
                nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
                uu, vv = np.meshgrid(np.linspace(0, 3, 25), np.linspace(0, 1, 5), indexing="xy")
                i = (np.arange(4)[:, None] * 25 + np.arange(24)[None, :]).reshape(-1)
                idx = np.stack([i, i + 1, i + 26, i, i + 26, i + 25], -1).reshape(-1)
                tan = np.concatenate([np.gradient(cen, axis=0) / np.linalg.norm(np.gradient(cen, axis=0), axis=-1, keepdims=True), np.ones((25, 1))], -1)
                tan = np.broadcast_to(tan[None], (5, 25, 4))
                mesh = (pos.reshape(-1, 3).astype(np.float32), nrm.reshape(-1, 3).astype(np.float32), np.stack([uu, vv], -1).reshape(-1, 2).astype(np.float32),
                        idx.astype(np.uint32), tan.reshape(-1, 4).astype(np.float32))
                # make sure faces point away from the sweep centre line
                p0, p1, p2 = (mesh[0][mesh[3].reshape(-1, 3)[:, k]] for k in range(3))
                gn = np.cross(p1 - p0, p2 - p0)
                if np.sum(gn * mesh[1][mesh[3].reshape(-1, 3)[:, 0]]) < 0:
                    mesh = mesh[:3] + (mesh[3].reshape(-1, 3)[:, ::-1].reshape(-1).copy(),) + mesh[4:]
                add(mesh, m_arch[(ci + storey) % 2])

    # hanging cloth (wavy, double-sided) along the upper gallery
    for k in range(6):
        x0 = -L / 2 + 2.5 + k * 3.6
        wave = lambda s, t, k=k: 0.15 * np.sin(s * 9 + k) * (0.3 + t) + 0.05 * np.sin(t * 14 + 2 * k)
        add(grid(40, 48, (x0, Hh / 2 + 0.2, -Wd / 2 + 1.9 if k % 2 == 0 else Wd / 2 - 1.9), (2.4, 0, 0), (0, 3.2, 0), height=wave), m_cloth[k % 3])

    # vases + metal ornaments + "lion head" reliefs
    vy = np.linspace(0, 1, 41)
    vr = 0.25 + 0.18 * np.sin(vy * 3.0 + 0.4) * (1 - 0.5 * vy)
    for k in range(8):
        add(revolve(vr, vy * 1.1, 48, u_scale=2), m_vase[k % 2], translate(-L / 2 + 3 + k * 2.6, 0.0, (-1) ** k * 1.4))
    for k in range(6):
        add(uv_sphere(0.22, 40, 20), m_metal[k % 2], translate(-L / 2 + 4 + k * 3.2, 2.3, (-1) ** k * (Wd / 2 - 0.5)))
    kk = rng.normal(size=(5, 3))
    lion = lambda d: sum(0.05 / (1 + i) * np.sin((4 + 3 * i) * (d @ kk[i])) for i in range(5))
    for k in range(4):
        add(uv_sphere(0.6, 96, 48, displace=lion), m_lion, translate(-L / 2 + 0.3 if k < 2 else L / 2 - 0.3, 2.0 + 3.5 * (k % 2), 0.0) @ scale(0.5, 1.0, 1.0))

    # foliage: potted plants built from ALPHA_MASK cards, plus card "chains" hanging from the roof
    n_fol_tris = int(target_tris * foliage_frac)
    plants = 10
    per = max(8, n_fol_tris // 2 // plants // 1)
    per_cards = max(4, int(per * 0.85))
    for k in range(plants):
        cx = -L / 2 + 2.2 + k * (L - 4.4) / (plants - 1)
        cz = (-1) ** k * 2.6
        add(cards(rng, per_cards, (cx, 1.6, cz), (1.3, 1.8, 1.3), 0.16), m_leaf[k % 3])
        add(revolve(0.3 + 0.1 * vy, vy * 0.7, 24), m_vase[k % 2], translate(cx, 0.0, cz))
    chain_cards = max(4, per - per_cards)
    for k in range(plants):
        cx = -L / 2 + 3.0 + k * (L - 6.0) / (plants - 1)
        add(cards(rng, chain_cards, (cx, Hh - 2.0, 0.0), (0.15, 3.6, 0.15), 0.07, up_bias=0.05), m_chain)

    # top up to the target triangle count with extra subdivision of a ceiling ornament grid
    remaining = target_tris - budget["tris"]
    if remaining > 200:
        n = int(np.sqrt(remaining / 2.0 / 2.0))
        bump = lambda s, t: 0.06 * np.sin(s * 40) * np.sin(t * 12)
        add(grid(2 * n, n, (-L / 2 + 0.5, 0.02, 1.0), (L - 1.0, 0, 0), (0, 0, -2.0), uv_scale=(10, 1), height=bump), m_cloth[0])  # nave carpet

    sc.camera = Camera(eye=(-L / 2 + 1.6, 2.2, 0.6), center=(L / 2, 3.2, -0.4), up=(0, 1, 0), fov=60.0)
    return sc


def bistro_like(target_tris=3_800_000, tex_size=512, seed=SEED_BASE + 5, distinct=200):
    """C5 stand-in: a street of instanced clutter (~`distinct` prim-meshes, thousands of nodes)."""
    rng = np.random.default_rng(seed)
    sc = Scene("bistro_like")
    mats = []
    for i in range(24):
        tint = tuple(0.3 + 0.6 * rng.random(3))
        mats.append(sc.add_material(pbrBaseColorTexture=sc.add_texture(tex_albedo(rng, tex_size, tint, pattern=["noise", "brick", "checker"][i % 3])),
                                    pbrMetallicFactor=float(i % 5 == 0), pbrRoughnessFactor=0.3 + 0.6 * rng.random()))
    leaf = sc.add_material(pbrBaseColorTexture=sc.add_texture(tex_leaf(rng, tex_size)), alphaMode=hd.ALPHA_MASK, doubleSided=1, pbrMetallicFactor=0.0)
    _add(sc, grid(200, 200, (-60, 0, 60), (120, 0, 0), (0, 0, -120), uv_scale=(60, 60)), mats[0])
    budget = 80_000
    pms = []
    tris_each = []
    for i in range(distinct):
        kind = i % 4
        if kind == 0:
            mesh = uv_sphere(0.5, 48, 24)
        elif kind == 1:
            y = np.linspace(0, 1, 33)
            mesh = revolve(0.2 + 0.15 * np.sin(y * (2 + i % 5)), y * (0.8 + 0.1 * (i % 7)), 48)
        elif kind == 2:
            mesh = box((0.8, 0.6 + 0.1 * (i % 5), 0.8), sub=12)
        else:
            mesh = cards(rng, 600, (0, 0.8, 0), (1.0, 1.4, 1.0), 0.12)
        pos, nrm, uv, idx, tan = mesh
        pms.append(sc.add_prim_mesh(pos, nrm, uv, idx, leaf if kind == 3 else mats[1 + i % 23], tangents=tan))
        tris_each.append(len(idx) // 3)
    i = 0
    while budget < target_tris:
        k = i % distinct
        x, z = (rng.random(2) - 0.5) * 110
        s = 0.6 + 1.6 * rng.random()
        sc.add_node(pms[k], translate(x, 0.0, z) @ rotate_y(rng.random() * 6.28) @ scale(s, s * (0.8 + 0.6 * rng.random()), s))
        budget += tris_each[k]
        i += 1
    sc.camera = Camera(eye=(-40, 6, 30), center=(10, 1, -10), up=(0, 1, 0), fov=55.0)
    return sc


def fuzz_scene(seed):
    """Random geometry chosen to stress the traversal: axis-aligned boxes on an integer lattice (rays parallel to faces, origins
    on box planes), slivers, tiny and huge triangles, coincident triangles (ties in t), mirrored / scaled instances, a mix of
    opaque, MASK and BLEND materials with NEAREST / LINEAR, REPEAT / MIRROR / CLAMP textures of non-power-of-two size."""
    from .scene import Scene, Camera, translate, scale, rotate_y
    from . import host_device as hd
    rng = np.random.default_rng(seed)
    sc = Scene(f"fuzz{seed}")
    texs = []
    for k in range(4):
        w, h = [(16, 16), (12, 20), (7, 5), (32, 8)][k]
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        img[..., 3] = np.where(rng.random((h, w)) < 0.4, 0, np.where(rng.random((h, w)) < 0.5, 255, img[..., 3]))
        texs.append(sc.add_texture(img, magFilter=int(rng.integers(0, 2)), wrapS=int(rng.integers(0, 3)), wrapT=int(rng.integers(0, 3))))
    mats = [sc.add_material(pbrBaseColorFactor=(0.8, 0.7, 0.6, 1.0), pbrMetallicFactor=0.0, pbrRoughnessFactor=0.8),
            sc.add_material(pbrBaseColorFactor=(0.5, 0.6, 0.9, 1.0), pbrMetallicFactor=0.3, pbrRoughnessFactor=0.4, doubleSided=1)]
    for k in range(4):
        mats.append(sc.add_material(pbrBaseColorTexture=texs[k], alphaMode=hd.ALPHA_MASK if k % 2 == 0 else hd.ALPHA_BLEND, alphaCutoff=float(rng.choice([0.0, 0.3, 0.5])),
                                    pbrBaseColorFactor=(1.0, 1.0, 1.0, float(rng.choice([1.0, 0.7]))), doubleSided=int(k < 2), pbrMetallicFactor=0.0))
    # lattice of unit boxes (shared planes -> ties, rays along faces)
    bpos, bnrm, buv, bidx, btan = box((1, 1, 1))
    bm = sc.add_prim_mesh(bpos, bnrm, buv, bidx, mats[0], tangents=btan)
    for _ in range(12):
        c = rng.integers(-3, 4, 3).astype(np.float32)
        sc.add_node(bm, translate(*c))
    sc.add_node(bm, translate(0, -5.5, 0) @ scale(40.0, 1.0, 40.0))                      # huge floor
    sc.add_node(bm, translate(2, 0, 2) @ scale(-1.0, 1.0, 1.0))                          # mirrored
    sc.add_node(bm, translate(-2, 1, 1) @ rotate_y(0.3) @ scale(0.01, 3.0, 0.01))        # sliver-like column
    # random soup with alpha materials, incl. duplicated (coincident) triangles
    for mi in range(1, len(mats)):
        n = 60
        p = rng.uniform(-4, 4, (n, 1, 3)) + rng.normal(0, 0.8, (n, 3, 3))
        p = np.concatenate([p, p[:6]], 0).reshape(-1, 3).astype(np.float32)            # 6 coincident copies
        tri = np.arange(len(p), dtype=np.uint32)
        nn = np.cross(p[1::3] - p[0::3], p[2::3] - p[0::3]); nn /= np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-20)
        pm = sc.add_prim_mesh(p, np.repeat(nn, 3, 0), rng.uniform(-2, 3, (len(p), 2)).astype(np.float32), tri, mats[mi])
        sc.add_node(pm)
    eye = rng.integers(-3, 4, 3).astype(np.float32) + np.float32(0.5) * (seed % 2)     # sometimes exactly on lattice planes
    eye[2] = 9.0
    sc.camera = Camera(eye=tuple(eye), center=(float(eye[0]), float(eye[1]), 0.0) if seed % 3 == 0 else (0.0, 0.0, 0.0), up=(0, 1, 0), fov=50.0)
    return sc
