"""Phase timeline of the ping-pong bf16x6 backward kernel (dAdj tiles; wave 0 = group A, wave 4 = group B)."""
import sys, os, torch
sys.path.insert(0, '.')
os.environ["MARIUS_SCORES"] = "b"
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
for w, nm, labels in ((0, "group A", ["compute", "barrier", "prep", "barrier"]), (1, "group B", ["prep+stage", "barrier", "compute", "barrier"])):
    acc = {l: [] for l in labels}
    pro, tot = [], []
    for wg in range(256):
        x = b[wg, w]
        n = int((x != 0).sum())
        if n < 2 + 4 * 6: continue
        d = np.diff(x[:n])
        pro.append(d[0])
        for t in range(1, 6):
            for i, l in enumerate(labels):
                acc[l].append(d[1 + 4 * t + i])
        tot.append(x[n - 1] - x[0])
    print(nm, "samples", len(tot), "prologue %.0f total %.0f" % (np.mean(pro), np.mean(tot)))
    for l in labels:
        v = np.array(acc[l]); print("   %-12s mean %7.0f p50 %7.0f p90 %7.0f" % (l, v.mean(), np.median(v), np.percentile(v, 90)))
