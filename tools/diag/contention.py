"""The ViT-B/16 training step while another stream holds CUs (a stand-in for RCCL kernels during an overlapped all-reduce):
ms/step alone and with `n_wg` hog workgroups (one CU each) busy `duty` of the time in bursts of `burst_us`.
python tools/diag/contention.py [n_wg] [burst_us] [duty]      (build: hipcc -shared -fPIC --offload-arch=gfx950 tools/diag/cu_hog.hip -o diag_bin/libcuhog.so)"""
import ctypes
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vit_amd import dp, ops  # noqa: E402
from vit_amd.models import ViT  # noqa: E402

n_wg = int(sys.argv[1]) if len(sys.argv) > 1 else 32
burst_us = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
duty = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
hog = ctypes.CDLL(os.path.join(ROOT, "diag_bin", "libcuhog.so"))
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
c = bench.CONFIGS["vit_b16"]
torch.manual_seed(0)
model = ViT(bench.make_config(c)).to(dev)
trainer = dp.Trainer(model, lr=1e-4, weight_decay=0.05)
x = torch.randn(c["batch"], *c["chw"], device=dev)
y = torch.randint(0, c["out_d"], (c["batch"],), device=dev)


def step():
	trainer.zero_grad()
	with torch.autocast("cuda", dtype=torch.bfloat16):
		loss = ops.cross_entropy(model(x), y)
	loss.backward()
	trainer.step()


def timed(n=15):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(n):
		step()
	torch.cuda.synchronize()
	return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
	step()
alone = timed()
side = torch.cuda.Stream()
stop = False


def hog_loop():
	torch.cuda.set_device(dev)
	ticks = int(burst_us * 100)  # 100 MHz wall clock
	period = burst_us / duty * 1e-6
	nxt = time.perf_counter()
	k = 0
	while not stop:
		assert hog.hog_launch(n_wg, ticks, 96 * 1024, side.cuda_stream) == 0
		k += 1
		if k % 8 == 0:
			side.synchronize()
		nxt += period
		d = nxt - time.perf_counter()
		if d > 0:
			time.sleep(d)
		else:
			nxt = time.perf_counter()


th = threading.Thread(target=hog_loop)
th.start()
time.sleep(0.05)
contended = timed()
stop = True
th.join()
side.synchronize()
again = timed()
ideal = alone * (1 + duty * n_wg / 256.0)
print(f"{os.environ.get('VITK_P8_EVEN_ROUNDS', 'unset')=}, {os.environ.get('VITK_WGRAD_OVERLAP', '1')=}: alone {alone:.2f} ms/step, with {n_wg} CUs held {duty:.0%} of the time in {burst_us:.0f} us bursts {contended:.2f} ms/step "
	f"(+{100 * (contended / alone - 1):.1f} %; losing exactly the held CU-time would be {ideal:.2f}), alone again {again:.2f}", flush=True)
