"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU ASAN is not available on this pool).
The checker everything else is compared against must not itself read or write outside its planes on the edge shapes the
parity tests use (W < max_disp, W = 1, max_disp = 1, empty / ragged masks).  Both rounding variants (fmaf chain / mul+add)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fma", [1, 0])
def test_oracle_c_is_clean_under_asan_ubsan(tmp_path, fma):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_sanitized")
    cmd = [gcc, "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-ffp-contract=off", "-fopenmp", "-DORACLE_FMA=%d" % fma, os.path.join(ROOT, "tests", "oracle_sanitize_driver.c"),
           os.path.join(ROOT, "oracle", "spamat_oracle.c"), "-o", exe, "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("this gcc has no sanitizer runtime: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "SANITIZED_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
