"""ms per step in consecutive groups of five steps of the default workload (C++ SynchronousTrainer), with the allocator's device-allocation count beside each
group: is there a warm-up tail inside the driver's 20-step window?  (round 5: 0.579 -> 0.557 ms over the first 30 steps with no allocation after the first
group — a clock ramp, not work; one 0.4 ms blip around step 37.)   python tools/step_times.py"""
import sys, time, math, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as Bm
import marius_amd
from marius_amd import hip as H
H.lib()
dev = torch.device('cuda:0')
cfg = Bm.WORKLOADS['freebase86m']
num_nodes, R, d, B, C, N = cfg['num_nodes'], cfg['num_relations'], cfg['d'], cfg['B'], cfg['C'], cfg['N']
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
# groups of 5 steps, timed with a sync at both ends (as the bench does for its whole region)
out = []
for g in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.train_steps(5)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 5 * 1e3)
    st = torch.cuda.memory_stats()
    print("group %d: %.3f ms/step  device allocs so far %d, reserved %.2f GB" % (g, out[-1], st.get("num_device_alloc", 0), torch.cuda.memory_reserved() / 1e9))
print("ms per step in consecutive groups of 5 steps:", ["%.3f" % x for x in out])
print("allocator: reserved %.1f GB, num_alloc_retries %d" % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_stats().get("num_alloc_retries", 0)))
