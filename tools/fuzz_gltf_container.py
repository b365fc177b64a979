"""CPU: corrupts valid files (truncation, byte flips) and feeds them to the C++ glTF importer built with AddressSanitizer + UBSan:
  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude tools/asan_gltf_main.cpp \
      vk_raytrace_amd/csrc/pt_gltf.cpp vk_raytrace_amd/csrc/pt_host.cpp -lz -o /tmp/asan_gltf
Every file must be either decoded or rejected with a message; a sanitizer report is a bug."""
import sys, os, subprocess, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np
from vk_raytrace_amd import gltf, synth
rng = np.random.default_rng(7)
d = tempfile.mkdtemp()
base = os.path.join(d, "base.glb"); gltf.save_gltf(synth.feature_box(tex_size=8, lights=True), base)
data = open(base, "rb").read()
paths = []
for t in range(400):
    x = bytearray(data)
    m = t % 4
    if m == 0: x = x[: rng.integers(12, len(x))]
    elif m == 1:
        for _ in range(rng.integers(1, 8)): x[rng.integers(12, len(x))] = rng.integers(0, 256)
    elif m == 2:   # corrupt inside the JSON chunk (first ~40 KB)
        for _ in range(rng.integers(1, 5)): x[rng.integers(20, min(len(x), 20000))] = rng.integers(32, 127)
    else:
        i = rng.integers(12, len(x) - 16); x[i:i+8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
    p = os.path.join(d, f"f{t}.glb"); open(p, "wb").write(bytes(x)); paths.append(p)
a = subprocess.run(["/tmp/asan_gltf"] + paths, capture_output=True, text=True)
print("ASAN rc", a.returncode, a.stdout.strip(), a.stderr.strip()[:1500])
