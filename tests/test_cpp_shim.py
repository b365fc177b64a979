"""The header-only C++ Renderer shim (include/pt_renderer.hpp) compiles against the C ABI, links with
libptmi.so, and either renders (GPU box) or fails loudly with PT_ERR_NO_DEVICE (no CPU fallback)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "shim_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "renderer_shim_test.cpp"),
                           "-L", os.path.join(ROOT, "vk_raytrace_amd"), "-l:libptmi.so", "-Wl,-rpath," + os.path.join(ROOT, "vk_raytrace_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
    return exe


def test_cpp_shim_builds_and_fails_loudly_without_gpu(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("NO_DEVICE") or out.stdout.startswith("OK")


@pytest.mark.gpu
def test_cpp_shim_renders_on_gpu(tmp_path):
    from vk_raytrace_amd import gltf, synth
    glb = str(tmp_path / "quad.glb")
    gltf.save_gltf(synth.quad_scene(), glb)     # the same quad through Scene::load's replacement (pt_gltf_load via HipPathTracer::loadGltf)
    out = subprocess.run([_build(tmp_path), glb], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
