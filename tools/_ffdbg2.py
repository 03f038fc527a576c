import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch.nn.functional as F
from mofa_video_amd import lib, ops
import test_ff320_gpu as T
lib.load()
prm = T._params(3)
w1p, b1f, w2p, b2 = T._pack(prm)
w1, b1, w2, b2_, gamma, beta = [t.cuda().float() for t in prm]
M = 128
g = torch.Generator(device='cuda').manual_seed(M)
x = (torch.randn(M, 320, generator=g, device='cuda') * 1.3 + 0.2).half()
ref, _ = T._reference(x, prm)
got = ops.ff320(x, w1p, b1f, w2p, b2).float()
err = got - ref
xn = F.layer_norm(x.float(), (320,), gamma, beta, 1e-5)
p = xn @ w1.T + b1
h = p[:, :1280] * F.gelu(p[:, 1280:])             # [M, 1280]
contrib = torch.stack([h[:, 32*c:32*c+32] @ w2[:, 32*c:32*c+32].T for c in range(40)])   # [40, M, 320]
A = contrib.reshape(40, -1).T                      # [M*320, 40]
sol = torch.linalg.lstsq(A, err.reshape(-1, 1)).solution.flatten()
print("alpha per chunk:", [round(v, 2) for v in sol.tolist()])
print("residual after fit:", (A @ sol - err.reshape(-1)).norm().item(), "of", err.norm().item())
# per wave (32-row group) error norms
print("err norm per 32-row group:", [round(err[32*i:32*i+32].norm().item(), 3) for i in range(M // 32)])
# bias hypothesis: error explained by pre-activation bias shifts? try hidden-space: delta_h = err @ pinv(w2.T)
xf_ = x.float()
ffo = ref - xf_
def proj(u, v): return ((u * v).sum() / (v * v).sum()).item()
print("proj err on x:", proj(err, xf_), " on ff:", proj(err, ffo))
# is err a function of the output column block?  per 8-col piece norms (40 pieces)
print("err per piece:", [round(err[:, 8*p:8*p+8].norm().item(), 2) for p in range(40)])
# compare with residual taken from a different piece: got - (ffo + x_piece_shifted)
for sh in (-2, -1, 1, 2):
    xs_ = torch.roll(xf_.reshape(M, 40, 8), sh, dims=1).reshape(M, 320)
    print("shift", sh, "norm", (got - (ffo + xs_)).norm().item())
print("norm err", err.norm().item(), "norm ff", ffo.norm().item(), "norm x", xf_.norm().item())
# hidden-space check: recompute with the kernel's own fp16 roundings
xn16 = F.layer_norm(x.float(), (320,), None, None, 1e-5).half().float()
w1g = (w1 * gamma[None]).half().float()
b1f_ = b1 + w1 @ beta
p2 = xn16 @ w1g.T + b1f_
h2 = (p2[:, :1280] * F.gelu(p2[:, 1280:])).half().float()
y2 = (h2 @ w2.T + b2_).half().float() + xf_
print("vs emulated fp16 pipeline:", (got - y2).norm().item())
# gate/value swapped?
h3 = (p2[:, 1280:] * F.gelu(p2[:, :1280])).half().float()
print("vs swapped val/gate:", (got - ((h3 @ w2.T + b2_).half().float() + xf_)).norm().item())
# bias b1 missing?
p4 = xn16 @ w1g.T
h4 = (p4[:, :1280] * F.gelu(p4[:, 1280:])).half().float()
print("vs no b1:", (got - ((h4 @ w2.T + b2_).half().float() + xf_)).norm().item())
for name, pp in (("b1 val only", torch.cat([p4[:, :1280] + b1f_[:1280], p4[:, 1280:]], 1)), ("b1 gate only", torch.cat([p4[:, :1280], p4[:, 1280:] + b1f_[1280:]], 1)),
                 ("b1 swapped", torch.cat([p4[:, :1280] + b1f_[1280:], p4[:, 1280:] + b1f_[:1280]], 1))):
    hh = (pp[:, :1280] * F.gelu(pp[:, 1280:])).half().float()
    print("vs", name, (got - ((hh @ w2.T + b2_).half().float() + xf_)).norm().item())
