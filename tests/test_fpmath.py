"""The fp32 transcendental contract (include/pt_fpmath.h).

CPU part: the host evaluation is held to double-precision libm within the ulp bounds the header states, and to the IEEE special
cases.  GPU part: the gfx950 evaluation (pt_fpmath_eval through the C ABI) is BIT-IDENTICAL to the host evaluation -- this is what
makes path-traced frames of the product and of the oracle comparable bit for bit.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_probe = None


def probe():
    global _probe
    if _probe is None:
        out = os.path.join(tempfile.gettempdir(), f"pt_fpmath_probe_{os.getuid()}.so")
        src = os.path.join(ROOT, "tests", "cpp", "fpmath_probe.c")
        hdr = os.path.join(ROOT, "include", "pt_fpmath.h")
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            # same floating-point flags as the oracle and the HIP build: no contraction, no fast-math
            subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", out, src, "-lm"])
        _probe = C.CDLL(out)
    return _probe


def host_eval(name, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    if b is None:
        getattr(probe(), "probe_" + name)(C.c_void_p(a.ctypes.data), C.c_void_p(out.ctypes.data), C.c_long(a.size))
    else:
        b = np.ascontiguousarray(b, np.float32)
        getattr(probe(), "probe_" + name)(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(out.ctypes.data), C.c_long(a.size))
    return out


def max_ulp(got, ref64):
    ref32 = ref64.astype(np.float32)
    u = np.maximum(np.spacing(np.abs(ref32)).astype(np.float64), 2.0 ** -149)
    ok = np.isfinite(ref32) & (ref32 != 0)
    assert np.array_equal(np.isnan(got), np.isnan(ref32))
    assert np.isfinite(got[ok]).all()
    return float((np.abs(got[ok].astype(np.float64) - ref64[ok]) / u[ok]).max())


N = 400_000


def samples(kind, rng):
    if kind == "trig":
        return np.concatenate([rng.uniform(-10, 10, N), rng.uniform(-1e5, 1e5, N), rng.uniform(-1, 1, N) * 10.0 ** rng.uniform(-30, 0, N)]).astype(np.float32)
    if kind == "unit":
        return np.concatenate([rng.uniform(-1, 1, N), 1 - 10.0 ** rng.uniform(-8, 0, N), -1 + 10.0 ** rng.uniform(-8, 0, N),
                               rng.uniform(-1, 1, N) * 10.0 ** rng.uniform(-30, 0, N)]).astype(np.float32)
    if kind == "exp":
        return np.concatenate([rng.uniform(-103, 88.7, N), rng.uniform(-1, 1, N)]).astype(np.float32)
    if kind == "pos":
        return np.concatenate([10.0 ** rng.uniform(-45, 38.5, N), rng.uniform(0.5, 2, N)]).astype(np.float32)
    raise KeyError(kind)


BOUNDS = [("sin", "trig", np.sin, 1.6), ("cos", "trig", np.cos, 1.6), ("tan", "trig", np.tan, 3.5), ("asin", "unit", np.arcsin, 2.6),
          ("acos", "unit", np.arccos, 1.6), ("exp", "exp", np.exp, 1.1), ("log", "pos", np.log, 1.0)]


@pytest.mark.parametrize("name,kind,ref,bound", BOUNDS)
def test_unary_accuracy(name, kind, ref, bound):
    x = samples(kind, np.random.default_rng(7))
    with np.errstate(all="ignore"):
        e = max_ulp(host_eval(name, x), ref(x.astype(np.float64)))
    assert e <= bound, f"{name}: {e:.2f} ulp"


def test_atan2_accuracy():
    rng = np.random.default_rng(8)
    y = (rng.uniform(-1, 1, N) * 10.0 ** rng.uniform(-10, 10, N)).astype(np.float32)
    x = (rng.uniform(-1, 1, N) * 10.0 ** rng.uniform(-10, 10, N)).astype(np.float32)
    assert max_ulp(host_eval("atan2", y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64))) <= 3.0


def test_pow_accuracy():
    rng = np.random.default_rng(9)
    with np.errstate(all="ignore"):
        a = (10.0 ** rng.uniform(-6, 6, N)).astype(np.float32)
        b = rng.uniform(-12, 12, N).astype(np.float32)
        assert max_ulp(host_eval("pow", a, b), np.power(a.astype(np.float64), b.astype(np.float64))) <= 1.1
        a = rng.uniform(0, 1, N).astype(np.float32)
        for e in (2.2, 1 / 2.2, 5.0, 4.0, 2.0, 0.5, 1.5):  # the exponents the shaders use
            b = np.full(N, e, np.float32)
            assert max_ulp(host_eval("pow", a, b), np.power(a.astype(np.float64), b.astype(np.float64))) <= 1.1
        a = (10.0 ** rng.uniform(-38, 38, N)).astype(np.float32)
        b = rng.uniform(-3, 3, N).astype(np.float32)
        assert max_ulp(host_eval("pow", a, b), np.power(a.astype(np.float64), b.astype(np.float64))) <= 1.1


def test_special_values():
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    f = lambda n, *a: host_eval(n, *[np.array([v], np.float32) for v in a])[0]
    assert f("sin", 0.0) == 0 and np.signbit(f("sin", -0.0)) and np.isnan(f("sin", inf)) and np.isnan(f("cos", nan)) and f("cos", 0.0) == 1
    assert f("asin", 1.0) == np.float32(np.pi / 2) and f("acos", 1.0) == 0 and f("acos", -1.0) == np.float32(np.pi) and np.isnan(f("acos", 1.0000001))
    assert f("exp", 0.0) == 1 and f("exp", 89.0) == inf and f("exp", -200.0) == 0 and f("exp", -inf) == 0 and np.isnan(f("exp", nan))
    assert f("log", 1.0) == 0 and f("log", 0.0) == -inf and np.isnan(f("log", -1.0)) and f("log", inf) == inf
    assert f("atan2", 0.0, 1.0) == 0 and f("atan2", 0.0, -1.0) == np.float32(np.pi) and f("atan2", 1.0, 0.0) == np.float32(np.pi / 2)
    assert f("atan2", -1.0, 0.0) == -np.float32(np.pi / 2) and f("atan2", 0.0, 0.0) == 0
    assert f("pow", 0.0, 0.0) == 1 and f("pow", 0.0, 2.0) == 0 and f("pow", 0.0, -1.0) == inf and f("pow", 2.0, 10.0) == 1024
    assert f("pow", -2.0, 3.0) == -8 and f("pow", -2.0, 2.0) == 4 and np.isnan(f("pow", -2.0, 0.5)) and f("pow", 1.0, nan) == 1
    assert f("pow", 2.0, 200.0) == inf and f("pow", 2.0, -200.0) == 0 and f("pow", 0.5, inf) == 0 and f("pow", inf, -1.0) == 0
    # denormal results are produced exactly like any other value
    assert f("exp", -100.0) == np.float32(np.exp(-100.0)) and f("pow", 2.0, -140.0) == np.float32(2.0 ** -140)
    assert abs(float(f("log", 1e-42)) - np.log(float(np.float32(1e-42)))) < 1e-5


GPU_CASES = [("sin", "trig"), ("cos", "trig"), ("tan", "trig"), ("asin", "unit"), ("acos", "unit"), ("exp", "exp"), ("log", "pos")]


@pytest.mark.gpu
def test_device_evaluation_is_bit_identical_to_host():
    from vk_raytrace_amd import capi
    L = capi.lib()
    ctx = C.c_void_p()
    assert L.pt_create(0, C.byref(ctx)) == 0
    rng = np.random.default_rng(11)

    def dev(name, a, b=None):
        out = np.empty_like(a)
        rc = L.pt_fpmath_eval(ctx, capi.PT_FN[name], a.size, a.ctypes.data, None if b is None else b.ctypes.data, out.ctypes.data)
        assert rc == 0, L.pt_last_error(ctx)
        return out

    def same(h, d, what):
        # NaN payloads are not part of the contract; everything else is compared as bits (signed zeros, denormals, infinities)
        hn, dn = np.isnan(h), np.isnan(d)
        assert np.array_equal(hn, dn), what
        bad = np.count_nonzero(h.view(np.uint32)[~hn] != d.view(np.uint32)[~hn])
        assert bad == 0, f"{what}: {bad} of {h.size} differ"

    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, 88.7228, 88.73, -103.9, -104.1, 1e5, 99999.99, 1e6, 1e30], np.float32)
    try:
        for name, kind in GPU_CASES:
            x = np.concatenate([samples(kind, rng), special])
            h, d = host_eval(name, x), dev(name, x)
            same(h, d, name)
        y = np.concatenate([(rng.uniform(-1, 1, N) * 10.0 ** rng.uniform(-10, 10, N)).astype(np.float32), np.repeat(special, special.size)])
        x = np.concatenate([(rng.uniform(-1, 1, N) * 10.0 ** rng.uniform(-10, 10, N)).astype(np.float32), np.tile(special, special.size)])
        same(host_eval("atan2", y, x), dev("atan2", y, x), "atan2")
        a = np.concatenate([(10.0 ** rng.uniform(-38, 38, N)).astype(np.float32), rng.uniform(0, 1, N).astype(np.float32), np.repeat(special, special.size)])
        b = np.concatenate([rng.uniform(-12, 12, N).astype(np.float32), rng.choice(np.array([2.2, 1 / 2.2, 5, 4, 2, 0.5], np.float32), N), np.tile(special, special.size)])
        same(host_eval("pow", a, b), dev("pow", a, b), "pow")
    finally:
        L.pt_destroy(ctx)
