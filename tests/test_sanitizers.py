"""CPU sanitizer job (GPU sanitizers do not exist on this pool): AddressSanitizer + UndefinedBehaviorSanitizer
over (a) the oracle -- the pin tests and the numerics-contract tests run again on oracle/liboracle_san.so -- and (b) the
product's host-only unit csrc/cssm_model.cpp (descriptor validation, build_rec, flatten order, the PMMH loop) driven by
tests/cpp/san_model.cpp.  `make -C oracle SAN=1` builds both."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


@pytest.fixture(scope="module")
def san_build():
    if os.environ.get("CSSM_ORACLE_LIB"):
        pytest.skip("already inside the sanitizer job")
    r = subprocess.run(["make", "-C", ORACLE, "SAN=1", "-s"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return os.path.join(ORACLE, "liboracle_san.so"), os.path.join(ORACLE, "san_model")


def test_host_model_unit_is_clean_under_asan_and_ubsan(san_build):
    r = subprocess.run([san_build[1]], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "san_model: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_oracle_is_clean_under_asan_and_ubsan(san_build):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan.so beside this gcc")
    env = dict(os.environ, LD_PRELOAD=asan, CSSM_ORACLE_LIB=san_build[0], ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_pins.py"),
                        os.path.join(ROOT, "tests", "test_numerics_contract.py"), os.path.join(ROOT, "tests", "test_pmmh.py"), "-m", "not gpu"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
