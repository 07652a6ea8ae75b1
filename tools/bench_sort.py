"""sort_unique timing: hand-written radix sort vs the rocPRIM chain (MARIUS_SORT=rocprim), Freebase86m batch shape."""
import os, sys, torch
sys.path.insert(0, '.')
from marius_amd import hip as H
dev = torch.device('cuda:0')
for n, hi, bits in ((200000, 86054151, 27), (50000, 14824, 14)):
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(hi, (n,), generator=g).to(dev)
    um = H.UniqueMap(n, dev)
    for _ in range(5):
        um.run(ids, bits)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 50
    a.record()
    for _ in range(K):
        um.run(ids, bits)
    b.record()
    torch.cuda.synchronize()
    u, inv = torch.unique(ids.cpu(), return_inverse=True)
    U = int(um.count.item())
    ok = U == u.numel() and torch.equal(um.uniq[:U].cpu(), u) and torch.equal(um.inverse.cpu(), inv)
    print("n=%d bits=%d  %.1f us per call  (%s)  correct=%s" % (n, bits, a.elapsed_time(b) / K * 1e3, os.environ.get("MARIUS_SORT", "own"), ok))
