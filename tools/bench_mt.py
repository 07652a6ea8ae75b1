"""MT19937 device fill: workgroup size A/B (MARIUS_MT_THREADS = 64 / 128 / 256), 100,000 words = one getNegatives() request of the bench batch.
Every variant is checked bit for bit against the host generator (marius_mt19937_fill_host = ATen's stream)."""
import os, sys, torch
sys.path.insert(0, '.')
from marius_amd import hip as H
dev = torch.device('cuda:0')
n, K = 100000, 40
for T in ("256", "128", "64"):
    os.environ["MARIUS_MT_THREADS"] = T
    H.reload_env()
    want = H.Generator(42).fill_host(3 * n + 17)
    g = H.Generator(42).to_device(dev)
    got = torch.cat([g.fill_device(n), g.fill_device(17), g.fill_device(2 * n)])  # ragged request sizes: the block index carries across calls
    torch.cuda.synchronize()
    ok = torch.equal(got.cpu(), want)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    for _ in range(3):
        g.fill_device(n, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K):
        g.fill_device(n, out)
    b.record()
    torch.cuda.synchronize()
    print("threads=%s  %.1f us per %d words (%.3f us per 624-word block)  bit-exact=%s" % (T, a.elapsed_time(b) / K * 1e3, n, a.elapsed_time(b) / K * 1e3 / (n / 624), ok))
