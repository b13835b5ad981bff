"""pytest plugin (`-p tests.sanitize_plugin`; the interpreter must have been started with LD_PRELOAD=libasan.so): the oracle
is loaded from its AddressSanitizer + UndefinedBehaviorSanitizer build (oracle/Makefile, target `san`) and the host builds
of the device headers (tests/native/*.cpp via tests/native_flags.py) are compiled with the same instrumentation.  Used by
tests/test_sanitizers_cpu.py, or by hand:

    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
        python -m pytest -p tests.sanitize_plugin tests/test_oracle_golden.py -q
"""
import os
import subprocess

SAN_FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]


def pytest_configure(config):
    if "libasan" not in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("tests.sanitize_plugin: start python with LD_PRELOAD=$(gcc -print-file-name=libasan.so)")
    import oracle
    here = os.path.dirname(os.path.abspath(oracle.__file__))
    subprocess.check_call(["make", "-C", here, "-s", "san"])
    oracle._LIB_PATH = os.path.join(here, "_build", "libm3oracle_san.so")
    oracle.build = lambda force=False: oracle._LIB_PATH
    from tests import native_flags
    native_flags.EXTRA[:] = SAN_FLAGS


def pytest_report_header(config):
    import oracle
    return "sanitizers: oracle = %s; host builds + %s" % (oracle._LIB_PATH, " ".join(SAN_FLAGS))
