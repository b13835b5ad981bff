"""SURVEY.md section 5 "race detection / sanitizers": the oracle (oracle/*.c) and the host builds of the product's device
headers (tests/native/{planar,panda}_dyn_host.cpp, spline_fit_host.cpp: the same planar_dyn.hpp / panda_dyn.hpp /
spline_fit.hpp the kernels compile, hand-unrolled slot and lane indexing included, in their one-lane form) re-run their own
tests under AddressSanitizer + UndefinedBehaviorSanitizer (-fno-sanitize-recover=all: the first finding aborts).  The
instrumented libraries are loaded into a python started with LD_PRELOAD=libasan.so; tests/sanitize_plugin.py swaps them in.
A canary proves the set-up reports what it is there to find."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_device_dynamics_on_host.py", "tests/test_spline_fit.py", "tests/test_oracle_golden.py",
          "tests/test_oracle_panda.py", "tests/test_dynamics_physics.py", "tests/test_refshaped_cpu.py"]


def asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("gcc has no libasan.so here")
    return p


def san_env(asan):
    # (one OpenMP thread per pytest worker: the workers already fill the cores, and idle OpenMP threads spin)
    return dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
                UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="1", OMP_WAIT_POLICY="passive",
                PYTHONPATH=ROOT + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else ""))


def test_the_set_up_reports_a_heap_overflow_and_an_integer_overflow(tmp_path):
    asan = asan_runtime()
    sys.path.insert(0, ROOT)
    from tests.sanitize_plugin import SAN_FLAGS
    src = tmp_path / "canary.c"
    src.write_text("#include <stdlib.h>\n#include <limits.h>\n"
                   "int heap(int i) { int* p = malloc(4 * sizeof(int)); int v = p[i]; free(p); return v; }\n"
                   "int ovf(int x) { return x + INT_MAX; }\n")
    lib = str(tmp_path / "libcanary.so")
    subprocess.check_call(["gcc"] + SAN_FLAGS + ["-shared", "-fPIC", str(src), "-o", lib])
    # (the out-of-bounds load is reported by whichever instrumentation sees it first: UBSan's object-size check or ASan)
    for call, words in (("heap(4)", ("heap-buffer-overflow", "insufficient space")), ("ovf(1)", ("signed integer overflow",))):
        r = subprocess.run([sys.executable, "-c", "import ctypes; ctypes.CDLL(%r).%s" % (lib, call)], env=san_env(asan),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and any(w in r.stderr for w in words), (call, r.returncode, r.stderr[:500])


def test_oracle_and_device_headers_on_the_host_are_clean_under_asan_and_ubsan():
    asan = asan_runtime()
    workers = str(max(1, min(8, (os.cpu_count() or 2) - 1)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "tests.sanitize_plugin", "-x", "-m", "not gpu", "-n", workers,
                        "-p", "no:cacheprovider"] + SUITES, cwd=ROOT, env=san_env(asan), capture_output=True, text=True, timeout=3000)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "sanitizers: oracle = " in r.stdout and "libm3oracle_san.so" in r.stdout, tail      # (the plugin was active)
    assert " passed" in r.stdout and "failed" not in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
