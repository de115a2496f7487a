"""The work plan of the plane-fed weight-gradient GEMM (mirror_nerf_amd/csrc/mnrf_dwp.h) checked on the host: the header's
inline functions -- the same code the kernel and its launcher run -- are compiled with g++ into a small harness
(tests/csrc/dwp_plan_check.cpp) that verifies, over a thousand random plans incl. empty and tiny evaluations, that every
32-sample stage of every (job, evaluation) pair has exactly one owning workgroup, that the owners of a pair are consecutive
workgroups (what the finish kernel sums over), and that partial-tile slots never collide."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dwp_plan(tmp_path):
    exe = tmp_path / "dwp_plan_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "mirror_nerf_amd", "csrc"),
                    os.path.join(ROOT, "tests", "csrc", "dwp_plan_check.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok")
