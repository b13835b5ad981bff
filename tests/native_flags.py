"""Extra compiler flags for the host builds of the product's device headers (tests/native/*.cpp): empty in a normal run,
the sanitizer flags when pytest runs with `-p tests.sanitize_plugin` (tests/test_sanitizers_cpu.py)."""
EXTRA = []


def host_flags(base):
    if not EXTRA:
        return list(base)
    return [f for f in base if f not in ("-O2", "-O3")] + EXTRA
