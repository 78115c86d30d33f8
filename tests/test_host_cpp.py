"""Host C++ layer: the Entity-style op classes + tag-dispatched executor (graphflow_amd/host), driven by compiled
test programs in the style of the reference's tests/test_RisiContraction_18_gpu.cu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
REF = os.environ.get("GF_REFERENCE", "/root/reference")


@pytest.fixture(scope="module")
def bins(gf, oracle):
    subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)
    return os.path.join(CPP, "bin")


def test_op_headers_drop_into_the_real_reference_tree():
    """Compile + link our op classes against the reference's OWN containers (only where it is mounted)."""
    if not os.path.isdir(os.path.join(REF, "GraphFlow")):
        pytest.skip("reference not mounted")
    out = os.path.join(CPP, "bin", "dropin_reference_check")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(
        ["g++", "-std=c++11", "-w", "-I" + os.path.join(REF, "GraphFlow"), "-I" + os.path.join(ROOT, "include"),
         "-I" + os.path.join(ROOT, "graphflow_amd", "host"), os.path.join(CPP, "dropin_reference_check.cpp"), "-o", out,
         "-L" + os.path.join(ROOT, "graphflow_amd", "csrc"), "-lgf_hip", "-Wl,-rpath," + os.path.join(ROOT, "graphflow_amd", "csrc"),
         "-Wl,-rpath,/opt/rocm/lib"])


def test_inline_rand_stream_is_glibcs_rand_stream(bins):
    """The slice masks of the dropout towers are drawn from libc's rand() stream (the reference's order); gf::LibcRandom steps that
    stream inline (a million draws per 1024-sample step) and hands it back in step: tests/cpp/test_libc_random.cpp."""
    r = subprocess.run([os.path.join(bins, "test_libc_random")], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_host_op_fails_loudly_without_a_gpu(bins):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([os.path.join(bins, "test_RisiContraction_hip"), "4", "4"], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,K", [(16, 8, 18), (10, 5, 18), (8, 64, 18), (5, 3, 18), (12, 16, 18), (10, 5, 50), (7, 32, 50), (8, 4, 10), (10, 5, 4)])
def test_entity_style_op_parity_f64(bins, N, C, K):
    r = subprocess.run([os.path.join(bins, "test_RisiContraction_hip"), str(N), str(C), str(K)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout


@pytest.mark.gpu
def test_entity_style_op_parity_f32_containers(bins):
    r = subprocess.run([os.path.join(bins, "test_RisiContraction_hip_f32"), "16", "8", "18"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_entity_style_mixers_and_vertex_chain(bins):
    r = subprocess.run([os.path.join(bins, "test_Mixers_hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout


@pytest.mark.gpu
def test_entity_style_slice_dropout(bins):
    """RisiContraction_18_dropout_hip: same rand() mask as the reference after the same srand(), train/test modes."""
    r = subprocess.run([os.path.join(bins, "test_Dropout_hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout


@pytest.mark.gpu
def test_model_level_dropin_reproduces_reference_training(bins):
    """SMP_omega_hip (BatchLearn / Predict / Feature / save_model / load_model) on the toy molecules of the reference's
    tests/test_SMP_omega.cpp: same srand -> same initial weights -> the real reference's loss trajectory."""
    r = subprocess.run([os.path.join(bins, "test_SMP_omega_hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout


@pytest.mark.gpu
def test_physics_and_pairgraphs_dropins_reproduce_reference_training(bins):
    """SMP_omega_physics_hip / SMP_omega_pairgraphs_hip (+ beta, sigma) on the toy molecules of the reference's
    tests/test_SMP_omega_physics.cpp / test_SMP_omega_pairgraphs.cpp: same srand -> the real classes' loss trajectories."""
    r = subprocess.run([os.path.join(bins, "test_SMP_physics_hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout
