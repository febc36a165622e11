"""The checker must not lean on undefined behaviour: the oracle's own CPU tests (hand-derived vectors, golden vectors, the
geometry check, the second restatement, the reference's invariants, the discriminators) run against a build of
oracle/rcw_oracle.c under AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle san), in a child interpreter with
the sanitizer runtime preloaded.  Any report aborts the child (-fno-sanitize-recover, ASan's default)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_TESTS = ["tests/test_oracle_hand_derived.py", "tests/test_golden.py", "tests/test_oracle_geometry.py", "tests/test_pyref_vs_oracle.py",
                "tests/test_reference_invariants.py", "tests/test_discriminators.py"]


def test_oracle_cpu_tests_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no libasan.so here")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"], check=True)
    env = dict(os.environ, RCW_ORACLE_SANITIZED="1", LD_PRELOAD=asan, PYTHONDONTWRITEBYTECODE="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + ORACLE_TESTS,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-3000:] + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout, tail
    # the sanitized library really was the one loaded
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); from oracle import oracle as O; O.lib(32); O.lib(64); "
                            "print(open('/proc/self/maps').read().count('_build/san/librcw_oracle'))"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and int(probe.stdout.strip().splitlines()[-1]) > 0, probe.stdout + probe.stderr
