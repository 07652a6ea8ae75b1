"""Which torch op mis-sorts the synthetic edge list above ~100 M rows on this stack?  (found while diagnosing the out-of-core bench)"""
import torch
dev = torch.device("cuda", 0)
for n in (30_000_000, 60_000_000, 100_000_000, 260_000_000):
    g = torch.Generator(device=dev).manual_seed(1)
    key = torch.randint(256, (n,), generator=g, device=dev)
    val, order = torch.sort(key, stable=True)
    ok_vals = bool((val[1:] >= val[:-1]).all())
    ok_perm = bool((key[order] == val).all())
    uniq_perm = int(torch.bincount(order, minlength=n).max()) == 1
    pay = torch.stack([torch.arange(n, device=dev), key], 1)
    got = pay[order]
    ok_index = bool((got[:, 1] == key[order]).all()) and bool((got[:, 0] == order).all())
    a32 = torch.argsort(key.to(torch.int32), stable=True)
    ok32 = bool((key[a32][1:] >= key[a32][:-1]).all())
    print(n, "sort values ascending", ok_vals, "| key[order]==values", ok_perm, "| order is a permutation", uniq_perm, "| 2-D row index", ok_index, "| int32 argsort", ok32, flush=True)
    del key, val, order, pay, got, a32
