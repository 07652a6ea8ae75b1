"""Stress: sort_unique calls back to back on one stream while another stream keeps every CU busy with large-LDS matmul kernels.
The hand-written sort hands tile counts between resident workgroups (spin on flagged granules): this must never stall."""
import sys, time, torch
sys.path.insert(0, '.')
from marius_amd import hip as H
dev = torch.device('cuda:0')
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
a = torch.randn(8192, 8192, device=dev)
b = torch.randn(8192, 8192, device=dev)
s_mm, s_sort = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.Generator().manual_seed(1)
ids = [torch.randint(10_000_000, (200000,), generator=g).to(dev) for _ in range(4)]
rel = torch.randint(14824, (50000,), generator=g).to(dev)
um, um2 = H.UniqueMap(200000, dev), H.UniqueMap(50000, dev)
torch.cuda.synchronize()
t0 = time.time()
done = 0
while done < calls:
    with torch.cuda.stream(s_mm):
        for _ in range(4):
            c = a @ b
    with torch.cuda.stream(s_sort):
        for k in range(200):
            um.run(ids[k & 3], 24)
            um2.run(rel, 14)
        done += 200
    s_sort.synchronize()
    if done % 4000 == 0:
        print("calls", done, "%.1f s" % (time.time() - t0), flush=True)
torch.cuda.synchronize()
u = torch.unique(ids[(199) & 3].cpu())
print("ok", int(um.count.item()) == u.numel(), "%.1f s" % (time.time() - t0))
