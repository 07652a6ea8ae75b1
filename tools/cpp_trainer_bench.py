import sys, time, math, torch
sys.path.insert(0, '/root/repo')
import marius_amd, bench
M = marius_amd.host()
from marius_amd import hip as H
dev = torch.device('cuda:0')
cfg = bench.WORKLOADS['freebase86m']
num_nodes, R, d, B, C, N = cfg['num_nodes'], cfg['num_relations'], cfg['d'], cfg['B'], cfg['C'], cfg['N']
limit = math.sqrt(6.0/(num_nodes+d))
table = torch.empty((num_nodes, d), device=dev).uniform_(-limit, limit)
state = torch.zeros((num_nodes, d), device=dev)
edges_all = bench.synth_edges(num_nodes, R, cfg['num_edges'], 'zipf', dev)
gen = M.MariusGenerator(42)
loader = M.DataLoader(M.InMemory(edges_all), M.InMemory(table), M.InMemory(state), M.CorruptNodeNegativeSampler(C, N, 0.0, False, M.LocalFilterMode.DEG, gen), gen, B, True)
dec = M.ComplEx(R, d, dev, True, M.EdgeDecoderMethod.CORRUPT_NODE)
model = M.Model(dec, M.SoftmaxCrossEntropy('sum'), M.LinkPredictionReporter(), dev); model.setup_optimizers(0.1); model.sparse_lr=0.1
tr = M.SynchronousTrainer(loader, model)
loader.initializeBatches(True)
tr.train_steps(5); torch.cuda.synchronize()
t0=time.perf_counter(); tr.train_steps(30); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print('cpp fused ms/step', dt/30*1e3, 'pos edges/s', B*30/dt)
tr.fused_update=False
tr.train_steps(3); torch.cuda.synchronize()
t0=time.perf_counter(); tr.train_steps(20); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print('cpp granular ms/step', dt/20*1e3)
