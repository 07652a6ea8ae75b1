"""Default workload (freebase86m shape) on the C++ SynchronousTrainer: TRAIN_STEPS steps from a fresh table, the last 20 timed.  For rocprofv3 runs at two
step counts (per-kernel time of the LATE steps = difference of the totals): does a step get slower as the table trains?   TRAIN_STEPS=60 python tools/train_n.py"""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bm
import marius_amd
from marius_amd import hip as H
H.lib()
dev = torch.device('cuda:0')
cfg = Bm.WORKLOADS['freebase86m']
num_nodes, R, d, B, C, N = cfg['num_nodes'], cfg['num_relations'], cfg['d'], cfg['B'], cfg['C'], cfg['N']
steps = int(os.environ.get("TRAIN_STEPS", "60"))
limit = math.sqrt(6.0 / (num_nodes + d))
table = torch.empty((num_nodes, d), dtype=torch.float32, device=dev).uniform_(-limit, limit, generator=torch.Generator(device=dev).manual_seed(0))
state = torch.zeros((num_nodes, d), dtype=torch.float32, device=dev)
edges_all = Bm.synth_edges(num_nodes, R, cfg['num_edges'], 'zipf', dev)
M = marius_amd.host()
gen = M.MariusGenerator(42)
sampler = M.CorruptNodeNegativeSampler(C, N, 0.0, False, M.LocalFilterMode.DEG, gen)
loader = M.DataLoader(M.InMemory(edges_all), M.InMemory(table), M.InMemory(state), sampler, gen, B, True)
dec = M.ComplEx(R, d, dev, True, M.EdgeDecoderMethod.CORRUPT_NODE)
model = M.Model(dec, M.getLossFunction('SOFTMAX_CE', 'sum', 0.1), M.LinkPredictionReporter(), dev)
model.setup_optimizers(0.1); model.sparse_lr = 0.1
trainer = M.SynchronousTrainer(loader, model)
loader.initializeBatches(True)
trainer.train_steps(steps - 20)
torch.cuda.synchronize(); t0 = time.perf_counter()
trainer.train_steps(20)
torch.cuda.synchronize()
print("steps %d: last 20 at %.4f ms/step, loss %.1f, table absmax %.4f" % (steps, (time.perf_counter() - t0) / 20 * 1e3, float(model.loss[0].item()), float(model.range_state[0])))
