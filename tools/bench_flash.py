"""Times the decoder trio (forward / loss / backward) at the bench batch shape: flash path vs the materialised-score kernels."""
import sys, time
import torch
sys.path.insert(0, '.')
from marius_amd import hip as H

dev = torch.device('cuda:0')
B, C, N, d, U, R = 50000, 50, 1000, 100, 200000, 14824
g = torch.Generator().manual_seed(0)
emb = (torch.randn(U, d, generator=g) * 0.3).to(dev)
edges = torch.stack([torch.randint(U, (B,), generator=g), torch.randint(R, (B,), generator=g), torch.randint(U, (B,), generator=g)], 1).to(dev)
dn, sn = torch.randint(U, (C, N), generator=g).to(dev), torch.randint(U, (C, N), generator=g).to(dev)
rel, inv = (torch.randn(R, d, generator=g) * 0.3 + 1).to(dev), (torch.randn(R, d, generator=g) * 0.3 + 1).to(dev)
for name, flags in (("flash", H.LP_TRAIN_ONLY), ("materialised", 0)):
    W = H.LpWorkspace(1, 0, d, B, C, N, True, H.REDUCE_SUM, 3, True, dev, flags=flags)
    W.bind(emb, edges, dn, sn, rel, inv, absmax=torch.cat([H.table_absmax(emb), H.table_absmax(rel, inv)]) if flags else None)  # fp16 records, as every trainer packs them
    for _ in range(3):
        W.forward(); W.loss(); W.backward()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    K = 20
    tf = tl = tb = 0.0
    for _ in range(K):
        ev[0].record(); W.forward(); ev[1].record(); W.loss(); ev[2].record(); W.backward(); ev[3].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tl += ev[1].elapsed_time(ev[2]); tb += ev[2].elapsed_time(ev[3])
    print("%-13s flash=%d  forward %.3f ms  loss %.3f ms  backward %.3f ms  total %.3f ms  loss=%.4f  ws=%.0f MB" % (
        name, W.layout.flash, tf / K, tl / K, tb / K, (tf + tl + tb) / K, float(W.loss_values()[0]), W.layout.total_bytes / 1e6))
H.profile_enable(True)
W = H.LpWorkspace(1, 0, d, B, C, N, True, H.REDUCE_SUM, 3, True, dev, flags=H.LP_TRAIN_ONLY)
W.bind(emb, edges, dn, sn, rel, inv, absmax=torch.cat([H.table_absmax(emb), H.table_absmax(rel, inv)]))
H.profile_reset()
for _ in range(10):
    W.forward(); W.loss(); W.backward()
torch.cuda.synchronize()
print(H.profile_read())
