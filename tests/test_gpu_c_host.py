"""The C-ABI from a plain-C host: examples/solve_batch.c is compiled with gcc against include/mpcgpu.h and
libmpcgpu.so and run as its own process (no Python, no torch in that process: the system HIP runtime)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_solves_through_the_c_abi(tmp_path):
    pkg = os.path.join(ROOT, "trajtrack_mpcndqn_rlboost_amd")
    exe = str(tmp_path / "solve_batch")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "solve_batch.c"), "-o", exe, "-L", pkg, "-lmpcgpu",
                           f"-Wl,-rpath,{pkg}", "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.run([exe, "6"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("problem")]
    assert len(lines) == 6 and all("status 0" in ln for ln in lines)
    assert "wavefronts per SIMD" in out.stdout
    # the reference's call pattern: ONE problem per call -> the latency kernel; end to end well under a millisecond
    one = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    print("\n" + one.stdout)
    assert "latency kernel 4" in one.stdout
    call_ms = float(one.stdout.split("best of 10:")[1].split("ms")[0])
    assert call_ms < 5.0, call_ms      # a sanity bound, not a benchmark (typically 0.3 ms; tools/latency.py measures it)
