"""Host-side AddressSanitizer run of the C ABI's validation paths (CPU, no GPU): `make asan-run` builds the library with its
host code instrumented and a small driver (tools/abi_validation.cpp) that calls the entry points with malformed arguments;
every call must be rejected with a negative status and the sanitizer must stay silent."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_validation_paths_under_host_asan():
	if shutil.which("hipcc") is None:
		pytest.skip("no hipcc")
	r = subprocess.run(["make", "-C", os.path.join(ROOT, "vit_amd", "csrc"), "-j8", "asan-run"], capture_output=True, text=True, timeout=900)
	tail = "\n".join((r.stdout + "\n" + r.stderr).splitlines()[-30:])
	assert r.returncode == 0, tail
	assert "all calls rejected as expected" in r.stdout, tail
	assert "AddressSanitizer" not in r.stdout + r.stderr, tail
