"""Time marius_gather_rows / segment_adagrad_scatter on a Freebase86m-sized table (debug tool)."""
import sys, time, torch
sys.path.insert(0, '.')
from marius_amd import hip as H
dev = torch.device('cuda:0')
n_nodes, d = 86_054_151, 100
table = torch.empty((n_nodes, d), device=dev).uniform_(-0.01, 0.01)
g = torch.Generator(device=dev).manual_seed(1)
for U in (160_000, 200_000):
    ids = torch.sort(torch.randint(n_nodes, (U,), generator=g, device=dev))[0]
    out = torch.empty((U, d), device=dev)
    for _ in range(3): H.gather_rows(table, ids, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): H.gather_rows(table, ids, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("gather U=%d: %.1f us, read %.2f TB/s, read+write %.2f TB/s" % (U, ms * 1e3, U * d * 4 / ms / 1e9, 2 * U * d * 4 / ms / 1e9))
    # reference point: torch index_select
    for _ in range(3): torch.index_select(table, 0, ids, out=out)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): torch.index_select(table, 0, ids, out=out)
    e1.record(); torch.cuda.synchronize()
    print("   torch.index_select: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
# contiguous copy of the same size for scale
src = table[:200_000]
dst = torch.empty_like(src)
for _ in range(3): dst.copy_(src)
torch.cuda.synchronize(); e0.record()
for _ in range(20): dst.copy_(src)
e1.record(); torch.cuda.synchronize()
print("contiguous copy 80 MB: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
