import sys, torch
sys.path.insert(0, '.')
from marius_amd import hip as H
from marius_amd.lp_step import DeviceLinkPredictionStep
from oracle.cpu_step import CpuLinkPredictionStep
dev = torch.device('cuda:0')
num_nodes, R, d, B, C, N, E, seed = 1_000_000, 14824, 100, 50000, 50, 1000, 200000, 42
g = torch.Generator().manual_seed(3)
table = (torch.rand(num_nodes, d, generator=g) - 0.5) * 0.4
state = torch.rand(num_nodes, d, generator=g) * 0.01
edges_all = torch.stack([torch.randint(num_nodes, (E,), generator=g), torch.randint(R, (E,), generator=g), torch.randint(num_nodes, (E,), generator=g)], 1)
t_d, s_d = table.to(dev), state.to(dev)
cpu = CpuLinkPredictionStep("COMPLEX", table, state, R, B, C, N)
step = DeviceLinkPredictionStep("COMPLEX", num_nodes, R, d, B, C, N, seed=seed, device=dev, node_table=t_d, node_state=s_d)
torch.manual_seed(seed)
perm_ref = torch.randperm(E)
perm = step.gen.randperm_host(E)
e32 = edges_all.to(torch.int32).to(dev)
batch = edges_all[perm_ref[:B]]
want = cpu.step(batch)
edges = H.select_edges(e32, perm.to(dev), 0, B)
W = step.step(edges)
torch.cuda.synchronize()
for name, got, ref in [("rel", step.rel, cpu.rel), ("inv_rel", step.inv_rel, cpu.inv_rel), ("rel_sum", step.rel_sum, cpu.rel_sum), ("inv_rel_sum", step.inv_rel_sum, cpu.inv_rel_sum),
                       ("rel_grad", step.rel_grad, want["rel_grad"]), ("inv_rel_grad", step.inv_rel_grad, want["inv_rel_grad"])]:
    err = (got.cpu() - ref).abs()
    rows = (err.max(1).values > 1e-4 * ref.abs().max()).nonzero().flatten()
    print(name, "max err %.3e  max ref %.3e  bad rows %d" % (err.max().item(), ref.abs().max().item(), rows.numel()), rows[:10].tolist())
    if rows.numel():
        r = rows[0].item()
        cnt = (batch[:, 1] == r).sum().item()
        print("   row", r, "count in batch", cnt, "got", got[r, :4].tolist(), "ref", ref[r, :4].tolist())
# direct index_add of grel(1)
ig = torch.zeros(R, d, dtype=torch.float64).index_add_(0, batch[:, 1], W.grel(1)[:, :d].cpu().double())
print("index_add(grel1) vs oracle inv_rel_grad", (ig.float() - want["inv_rel_grad"]).abs().max().item())
ig0 = torch.zeros(R, d, dtype=torch.float64).index_add_(0, batch[:, 1], W.grel(0)[:, :d].cpu().double())
print("index_add(grel0) vs oracle rel_grad", (ig0.float() - want["rel_grad"]).abs().max().item())
print("segsum(inv) vs index_add", (step.inv_rel_grad.cpu().double() - ig).abs().max().item(), " segsum(rel) vs index_add", (step.rel_grad.cpu().double() - ig0).abs().max().item())
