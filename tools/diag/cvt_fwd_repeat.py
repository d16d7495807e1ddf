# Diagnostic: forward passes of the small CvT, same model, repeated; which block's output changes first?
import sys, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_gpu_dp as T
from vit_amd import ops
gpu = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
x = torch.randn(8, 3, 32, 32, generator=g).to(gpu)
def poison():
    for n in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 20):
        t = torch.empty(n, dtype=torch.float32, device=gpu); t.fill_(float("nan")); del t
m = T._cvt(gpu)
caps = {}
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        caps.setdefault(name, []).append(o.detach().clone())
    return f
for n, mod in m.named_modules():
    if n.endswith("_embedding") or (n.count(".") == 1 and "blocks" in n):
        mod.register_forward_hook(hook(n))
grad = len(sys.argv) > 1 and sys.argv[1] == "grad"
for it in range(10):
    poison()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        if grad:
            out = m(x); out.float().sum().backward()
        else:
            with torch.no_grad(): out = m(x)
    caps.setdefault("logits", []).append(out.detach().clone())
for n, v in caps.items():
    print(n, "repeatable" if all(torch.equal(t, v[0]) for t in v) else "DIFFERS in runs %s" % [i for i, t in enumerate(v) if not torch.equal(t, v[0])])
v = caps["stage1_embedding"]
for i, t in enumerate(v):
    if not torch.equal(t, v[0]):
        d = (t.float() - v[0].float())
        nz = (d != 0).nonzero()
        print("run", i, "differing elements", nz.shape[0], "of", t.numel(), "max abs", d.abs().max().item(), "finite", bool(torch.isfinite(t).all()), "first idx", nz[:4].tolist(), "vals", [(t[tuple(j)].item(), v[0][tuple(j)].item()) for j in nz[:4].tolist()])
        break
