"""CPU: the oracle's own tests once more against its AddressSanitizer + UndefinedBehaviorSanitizer build
(`make -C oracle asan-test`): no out-of-bounds access, no signed overflow, no misaligned or invalid shift in the
restatement that every GPU result is checked against."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_under_asan_and_ubsan():
    if os.environ.get("VFE_ORACLE_LIB"):
        return  # this IS the sanitizer run: do not recurse
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan-test"], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
