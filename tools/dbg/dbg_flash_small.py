import sys, torch
sys.path.insert(0, '.')
from marius_amd import hip as H
from oracle import lp_oracle as O
from tests.test_gpu_flash import make_batch, run_flash
dev = torch.device('cuda:0')
B, C, N, d = 2, 4, 64, 100
U, R = 40, 11
dec = "DISTMULT"
emb, edges, dn, sn, rel, inv = make_batch(dec, B, C, N, d, U, R, seed=B + d)
W = run_flash(H, dev, dec, emb, edges, dn, sn, rel, inv, True)
# per-occurrence oracle in fp64
occ_ids = torch.cat([edges[:, 0], edges[:, 2], sn.flatten(), dn.flatten()])
L = occ_ids.numel()
emb_occ = emb[occ_ids].double()
e2 = torch.stack([torch.arange(B), edges[:, 1], torch.arange(B) + B], 1)
sn2 = (torch.arange(C * N) + 2 * B).reshape(C, N)
dn2 = (torch.arange(C * N) + 2 * B + C * N).reshape(C, N)
w = O.train_batch(dec, emb_occ, torch.zeros(L, d, dtype=torch.float64), e2, dn2, sn2, rel.double(), inv.double())
g = W.gocc()[:, :d].cpu().double()
err = (g - w["node_grad"]).abs().max(1).values
print("max per-occ err", err.max().item(), "at occ", err.argmax().item(), "of", L, " (2B=%d, CN=%d)" % (2 * B, C * N))
bad = (err > 1e-7).nonzero().flatten()
print("bad occs:", bad.tolist()[:40])
for o in bad[:6].tolist():
    print(o, "id", occ_ids[o].item(), "got", g[o, :4].tolist(), "want", w["node_grad"][o, :4].tolist())
print("dadj rows:", W.dadj(0).cpu()[:, :3], W.dadj(1).cpu()[:, :3])
print("lse", W.lse(0).cpu(), W.lse(1).cpu(), "dpos", W._view(W.layout.dpos[0], (W.layout.Bp,)).cpu())
