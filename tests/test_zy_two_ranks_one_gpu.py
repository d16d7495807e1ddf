"""Two data-parallel ranks, two processes, ONE GPU: the trainer's multi-rank control flow (replica broadcast, gradient
buckets reduced from the backward hooks while weight gradients run on the second stream, plain and pipelined optimizer
step) with a real reduction between processes.  tools/dp_two_ranks_one_gpu.py does the work and the checking (replicas
bit-identical after every configuration's steps, parameters within 2e-3 (bf16 steps) of a single-process run over the
whole batch); this test starts it under the launcher and reads its verdict."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_ranks(backend):
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	env = dict(os.environ, STACKS_AFTER="150", WATCHDOG="120", BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
		os.path.join(ROOT, "tools", "dp_two_ranks_one_gpu.py")]
	r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
	tail = "\n".join((r.stdout + "\n" + r.stderr).splitlines()[-40:])
	assert r.returncode == 0, tail
	lines = r.stdout.splitlines()
	assert "OK" in lines, tail
	# plain and pipelined optimizer step, each without and with LayerScale
	assert sum(1 for ln in lines if ln.startswith("pipelined_step=") and "replicas differ by 0.0e+00" in ln) == 4, tail


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_stay_identical_and_follow_the_single_process_run():
	_two_ranks("gloo")


@pytest.mark.gpu
def test_two_ranks_on_two_gpus_over_rccl():
	"""The same run with one GPU per rank over torch.distributed "nccl" (RCCL): the collectives as dp.GradSync issues them
	(asynchronous, on RCCL's stream, overlapping the backward and the second stream), no host staging.  Needs two GPUs;
	device_count() does not initialise the GPU in this process."""
	import torch
	if torch.cuda.device_count() < 2:
		pytest.skip("needs two GPUs")
	_two_ranks("nccl")
