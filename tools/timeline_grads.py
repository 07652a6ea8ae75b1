"""Per-phase cycle breakdown of the dAdj half of lp_grad16_kernel from cycle stamps (debug tool)."""
import sys, os, torch
sys.path.insert(0, '.')
import numpy as np
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
os.environ["MARIUS_TIMELINE_GRADS"] = "1"
st.step(edges_all[150000:200000].long().contiguous())
torch.cuda.synchronize()
H.lib().marius_debug_set_timeline(None)
b = buf.cpu().view(256, 2, 64).numpy().astype(np.int64)
names = ['compute(0)', 'write', 'issue', 'barrier', 'compute(1)+write+issue+barrier']
acc = {n: [] for n in names}
pro, tail, tot = [], [], []
for wg in range(256):
    for w in range(2):
        x = b[wg, w]
        n = int((x != 0).sum())
        if n < 2 + 5 * 6 + 2: continue
        d = np.diff(x[:n])
        pro.append(d[0])
        for t in range(6):
            for i, nm in enumerate(names):
                acc[nm].append(d[1 + 5 * t + i])
        tail.append(d[n - 2])
        tot.append(x[n - 1] - x[0])
for nm in names:
    v = np.array(acc[nm]); print('%-34s mean %8.0f  p50 %8.0f  p90 %8.0f cycles' % (nm, v.mean(), np.median(v), np.percentile(v, 90)))
print('prologue mean %.0f  epilogue mean %.0f  per-WG total mean %.0f  samples %d' % (np.mean(pro), np.mean(tail), np.mean(tot), len(tot)))
