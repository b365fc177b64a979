"""CPU: corrupts valid files (truncation, byte flips) and feeds them to the C++ glTF importer built with AddressSanitizer + UBSan:
  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude tools/asan_gltf_main.cpp \
      vk_raytrace_amd/csrc/pt_gltf.cpp vk_raytrace_amd/csrc/pt_host.cpp -lz -o /tmp/asan_gltf
Every file must be either decoded or rejected with a message; a sanitizer report is a bug."""
import sys, io, os, json, base64, subprocess, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np
from PIL import Image
from tests.test_gltf import _tri_doc
rng = np.random.default_rng(1)
img = rng.integers(0, 256, (33, 47, 3), dtype=np.uint8)
srcs = {}
for name, kw in {"base.jpg": dict(format="JPEG", quality=80), "prog.jpg": dict(format="JPEG", quality=80, progressive=True), "rst.jpg": dict(format="JPEG", quality=80, restart_marker_rows=1),
                 "a.png": dict(format="PNG")}.items():
    b = io.BytesIO(); Image.fromarray(img, "RGB").save(b, **kw); srcs[name] = b.getvalue()
d = tempfile.mkdtemp()
worker = r'''
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
from vk_raytrace_amd import capi
L = capi.lib()
ok = bad = 0
for path in sys.argv[1:]:
    h = C.c_void_p(); err = C.create_string_buffer(256)
    rc = L.pt_gltf_load(path.encode(), C.byref(h), err, 256)
    if rc == 0: ok += 1; L.pt_gltf_free(h)
    else: bad += 1
print("ok", ok, "rejected", bad)
'''
paths = []
for name, data in srcs.items():
    for trial in range(60):
        mode = trial % 3
        x = bytearray(data)
        if mode == 0: x = x[: rng.integers(4, len(x))]
        elif mode == 1:
            for _ in range(rng.integers(1, 6)): x[rng.integers(2, len(x))] = rng.integers(0, 256)
        else:
            i = rng.integers(2, len(x) - 8); x[i:i + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
        fn = f"{name}.{trial}" + os.path.splitext(name)[1]
        open(os.path.join(d, fn), "wb").write(bytes(x))
        doc = _tri_doc(); doc["images"] = [{"uri": fn}]; doc["textures"] = [{"source": 0}]
        doc["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
        gp = os.path.join(d, fn + ".gltf"); open(gp, "w").write(json.dumps(doc)); paths.append(gp)
out = subprocess.run([sys.executable, "-c", worker] + paths, capture_output=True, text=True)
print("rc", out.returncode, out.stdout.strip(), out.stderr.strip()[-300:]); import glob; files = sorted(glob.glob(os.path.join(d, "*.gltf"))); a = subprocess.run(["/tmp/asan_gltf"] + files, capture_output=True, text=True); print("ASAN rc", a.returncode, a.stdout.strip(), a.stderr.strip()[:1500])
