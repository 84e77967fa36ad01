"""CPU-only: the integer primitives of lattisense_amd/csrc/modarith.h (approximate-quotient Shoup product, sign-test conditional
subtraction, 128-bit multiply-accumulate) against unsigned __int128 arithmetic on 10^7 random and edge operands
(tests/cpp/test_modarith.cpp, built with g++ -fsanitize=undefined).  The device build runs the same formulas."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_modarith_primitives(tmp_path):
    exe = str(tmp_path / "test_modarith")
    subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-Wall", "-DLSA_EMULATE", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(ROOT, "tests", "cpp", "test_modarith.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK modarith" in out.stdout
