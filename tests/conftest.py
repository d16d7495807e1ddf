import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
	"""Return (dict of torch tensors, meta dict) for tests/golden/<name>.npz."""
	z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
	meta = json.loads(bytes(z["meta"]).decode())
	out = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
	return out, meta


def sub(d, prefix):
	"""Entries of d under 'prefix/' with the prefix stripped."""
	n = len(prefix) + 1
	return {k[n:]: v for k, v in d.items() if k.startswith(prefix + "/")}


# differences below this are rounding noise around an exact zero (e.g. the query gradient of a one-key softmax)
ABS_FLOOR = 1e-7


# With VITK_RECORD_ERRORS=<file> every relerr() value is kept (largest per test) and written out at the end of the session:
# the measured errors behind the documented tolerances.
_recorded = {}


def relerr(a, b):
	"""Normwise relative error max|a-b| / max|b| (the tolerance figure every parity test quotes)."""
	a = a.detach().double().cpu()
	b = b.detach().double().cpu()
	denom = b.abs().max().clamp_min(1e-30)
	err = (a - b).abs().max()
	val = 0.0 if err <= ABS_FLOOR else (err / denom).item()
	if os.environ.get("VITK_RECORD_ERRORS"):
		test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
		_recorded[test] = max(_recorded.get(test, 0.0), val if val == val else float("inf"))
	return val


def pytest_sessionfinish(session, exitstatus):
	path = os.environ.get("VITK_RECORD_ERRORS")
	if path and _recorded:
		os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
		with open(path, "w") as f:
			json.dump(_recorded, f, indent=0, sort_keys=True)


@pytest.fixture(scope="session")
def gpu():
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	return torch.device("cuda:0")
