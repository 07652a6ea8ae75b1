import os, sys, time, torch
sys.path.insert(0, '/root/repo'); os.chdir('/root/repo')
os.environ["MARIUS_FORCE_SHARDED"]="1"
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bench
from marius_amd import hip as H
from marius_amd.lp_step import DeviceLinkPredictionStep
from marius_amd.sharded import PipelinedShardedTrainer
cfg=dict(bench.WORKLOADS["freebase86m"]); num_nodes=cfg["num_nodes"]; d=100
table=torch.empty((num_nodes,d),device=dev).uniform_(-0.01,0.01); state=torch.zeros((num_nodes,d),device=dev)
edges_all=bench.synth_edges(num_nodes,cfg["num_relations"],cfg["num_edges"],"zipf",dev,seed=1)
st=DeviceLinkPredictionStep("COMPLEX",num_nodes,cfg["num_relations"],d,cfg["B"],cfg["C"],cfg["N"],seed=42,device=dev)
perm=st.gen.randperm_host(edges_all.size(0)).to(dev)
side=dist.new_group(backend="gloo")
tr=PipelinedShardedTrainer(st,table,state,edges_all,perm,0,1,num_nodes,sync_interval=16,side_group=side,staleness=1)
ts=[]
for i in range(20):
    torch.cuda.synchronize(); t0=time.perf_counter(); tr.step(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print(" ".join("%.2f"%t for t in ts))
dist.destroy_process_group()
