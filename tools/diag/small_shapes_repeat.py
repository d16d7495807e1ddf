# Diagnostic: repeat small-shape residual GEMMs and norm backward passes with residual gradients between allocations filled
# with NaN patterns: any read of memory the call did not write shows up as a changed or non-finite result.
import sys, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from conftest import relerr
from vit_amd import ops
import vit_amd._lib as L
gpu = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
def poison():
    for n in (1 << 12, 1 << 16, 1 << 20, 1 << 22):
        t = torch.empty(n, dtype=torch.float32, device=gpu); t.fill_(float("nan")); del t
for (M, N, K) in [(512, 32, 32), (512, 32, 64), (128, 64, 64), (128, 64, 128), (32, 128, 128), (8192, 16, 16), (8192, 16, 64), (2048, 32, 32), (2048, 32, 128), (512, 64, 64), (512, 64, 256), (8192, 64, 64), (50000, 48, 48)]:
    x = torch.randn(M, K, generator=g).to(gpu).bfloat16(); w = (torch.randn(N, K, generator=g) * 0.2).to(gpu).bfloat16()
    b = torch.randn(N, generator=g).to(gpu); res = torch.randn(M, N, generator=g).to(gpu).bfloat16()
    ref = res.double() + x.double() @ w.double().t() + b.double()
    outs = []
    for it in range(12):
        poison()
        y = ops.linear_fwd(x, w, b, residual=res)
        outs.append(y.clone()); del y
    same = all(torch.equal(o, outs[0]) for o in outs)
    print("gemm+residual", (M, N, K), L.load().vitk_gemm_last_kernel().decode(), "repeatable" if same else "DIFFERS", "err %.2e" % relerr(outs[0], ref), "finite", bool(torch.isfinite(outs[0]).all()))
for kind in ("rmsnorm", "layernorm"):
    for (rows, d) in [(8192, 16), (2048, 32), (512, 64), (8192, 64), (5000, 48), (3000, 192)]:
        x = torch.randn(rows, d, generator=g).to(gpu).bfloat16(); dy = torch.randn(rows, d, generator=g).to(gpu).bfloat16(); dres = torch.randn(rows, d, generator=g).to(gpu).bfloat16()
        wgt = (torch.randn(d, generator=g) * 0.1).to(gpu); bias = torch.zeros(d, device=gpu)
        y, _, mean, rstd = ops.norm_fwd(x, wgt, bias if kind == "layernorm" else None, kind, 1e-6, True)
        outs = []
        for it in range(12):
            poison()
            dx, dw, db = ops.norm_bwd(dy, x, wgt, mean, rstd, kind, True, dres=dres)
            outs.append((dx.clone(), dw.clone())); del dx, dw
        same = all(torch.equal(o[0], outs[0][0]) for o in outs)
        samew = max(relerr(o[1], outs[0][1].double()) for o in outs)
        print("norm_bwd+dres", kind, (rows, d), "dx repeatable" if same else "dx DIFFERS", "dw spread %.1e" % samew, "finite", bool(torch.isfinite(outs[0][0]).all()))

# the whole fused block, forward only, at the three stage geometries of the small CvT
from vit_amd.layers.cvt import CvTBlock
from vit_amd.layers import RMSNorm
class C: pass
for (dim, heads, hw, stride, ratio) in [(32, 1, 8, 1, 2.0), (64, 2, 4, 2, 2.0), (128, 2, 2, 2, 1.0)]:
    c = C(); c.d_model = dim; c.d_ff = int(dim * ratio); c.n_head = heads; c.norm = RMSNorm; c.activation = "silu"; c.dropout = 0.0; c.use_layer_scale = False
    torch.manual_seed(1)
    blk = CvTBlock(c, 3, stride, 1).to(gpu)
    x = torch.randn(8, hw * hw, dim, generator=g).to(gpu).bfloat16()
    outs = []
    for it in range(12):
        poison()
        with torch.no_grad():
            yy, _, _ = blk(x, hw, hw)
        outs.append(yy.clone()); del yy
    print("fused block fwd", (dim, heads, hw, stride), "repeatable" if all(torch.equal(o, outs[0]) for o in outs) else "DIFFERS", "finite", bool(torch.isfinite(outs[0]).all()))
