import sys; sys.path.insert(0,'/root/repo')
import torch
from vit_amd import ops
dev=torch.device('cuda:0')
for M,N in ((33280,192),(33280,768),(50432,768),(50432,3072)):
	x=torch.randn(M,N,device=dev).to(torch.bfloat16)
	for _ in range(3): ops.colsum(x)
	ts=[]
	for _ in range(10):
		e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
		torch.cuda.synchronize(); e0.record()
		for _ in range(20): ops.colsum(x)
		e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/20*1e3)
	t=min(ts); print(f"colsum {M}x{N}: {t:.1f} us {M*N*2/t/1e3:.0f} GB/s")
