"""Per-phase cycle breakdown of lp_scores_res_kernel from s_memtime stamps (debug tool)."""
import sys, math, torch
sys.path.insert(0, '.')
import os
os.environ["MARIUS_SCORES_IL"] = "0"
os.environ["MARIUS_SCORES_PS"] = "0"
os.environ.setdefault("MARIUS_SCORES", "a")
import bench
from marius_amd import hip as H
from marius_amd.lp_step import DeviceLinkPredictionStep
dev = torch.device('cuda:0')
cfg = bench.WORKLOADS['freebase86m']
nn = 5_000_000
table = torch.randn(nn, 100, device=dev) * 0.01
state = torch.zeros(nn, 100, device=dev)
st = DeviceLinkPredictionStep('COMPLEX', nn, cfg['num_relations'], 100, cfg['B'], cfg['C'], cfg['N'], device=dev, node_table=table, node_state=state)
edges_all = bench.synth_edges(nn, cfg['num_relations'], 1_000_000, 'zipf', dev)
for s in range(3):
    st.step(edges_all[s*50000:(s+1)*50000].long().contiguous())
buf = torch.zeros(256*2*64, dtype=torch.int64, device=dev)
H.lib().marius_debug_set_timeline(H.ptr(buf))
st.step(edges_all[150000:200000].long().contiguous())
torch.cuda.synchronize()
H.lib().marius_debug_set_timeline(None)
b = buf.cpu().view(256, 2, 64)
names = ['mfma_issue', 'lds_write(+wait loads)', 'issue_loads', 'stores+lse', 'barrier']
import numpy as np
acc = {n: [] for n in names}
tot = []
pro = []
life = []
for wg in range(256):
    for w in range(2):
        x = b[wg, w].numpy()
        n = int((x != 0).sum())
        if n < 6: continue
        d = np.diff(x[:n].astype(np.int64))
        # stamps: entry, after prologue, then per tile 5 stamps
        pro.append(d[0])
        ntile = (n - 2) // 5
        for t in range(ntile):
            for i, nm in enumerate(names):
                acc[nm].append(d[1 + t*5 + i])
        tot.append(x[n-1] - x[0])
        if w == 0: life.append((int(x[0]), int(x[n-1])))
for nm in names:
    v = np.array(acc[nm]); print('%-26s mean %8.0f  p50 %8.0f  p90 %8.0f cycles' % (nm, v.mean(), np.median(v), np.percentile(v, 90)))
print('per-WG total (4 tiles) mean', np.mean(tot), 'cycles; samples', len(tot))

print('prologue mean %.0f p50 %.0f p90 %.0f' % (np.mean(pro), np.median(pro), np.percentile(pro, 90)))
life.sort()
t0 = life[0][0]
print('first 256 WGs of XCD0: first start 0, last end %d cycles' % (max(e for s_, e in life) - t0))
ev = sorted([(s_ - t0, 1) for s_, e in life] + [(e - t0, -1) for s_, e in life])
cur = 0; prev = 0; hist = {}
for tm, dl in ev:
    hist[cur] = hist.get(cur, 0) + (tm - prev); prev = tm; cur += dl
tot_t = sum(hist.values())
print('concurrency histogram (active WGs of the sampled 256 : share of time):', {k_: round(v / tot_t, 3) for k_, v in sorted(hist.items()) if v / tot_t > 0.02})
print('starts (first 20, cycles):', [s_ - t0 for s_, e in life[:20]])
