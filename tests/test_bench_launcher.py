"""`python bench.py --gpus N` started plainly launches its N ranks itself and relays rank 0's line (CPU: gloo ranks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks_and_relays_one_line():
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
	env["OMP_NUM_THREADS"] = "1"
	p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env,
		stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
	assert p.returncode == 0, p.stderr[-2000:]
	lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
	assert len(lines) == 1, p.stdout
	out = json.loads(lines[0])
	assert out == {"selftest": True, "n_gpus": 2, "sum": 3.0}


def test_bench_relays_a_failing_rank_as_nonzero_exit():
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
	# without a GPU the ranks fail at torch.cuda.set_device: the launcher must report that, not swallow it
	p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
		stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
	import torch
	if torch.cuda.is_available():
		return
	assert p.returncode != 0
	assert p.stdout.strip() == ""
