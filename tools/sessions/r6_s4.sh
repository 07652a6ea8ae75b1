#!/bin/bash
# round 6 session 4: the fused map launch under a watchdog (bounded waits + give-up records): phases bisected with one workgroup, then by workgroup count
tag=${1:-r6s4}
ulimit -c 0
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0))"
for k in 1 2 3 4 6 8 9 10 11 12; do
  echo "== one workgroup, first $k phases"
  MARIUS_PM_MAXPH=$k MARIUS_PM_NWG=1 PM_TRIALS=1 timeout -s KILL 90 python -u tools/pm_debug.py 2>&1 | grep -v amdgpu.ids | tee $out/pm_maxph$k.txt | cut -c1-300 | tail -8
done
for n in 2 8 0; do
  echo "== MARIUS_PM_NWG=$n"
  MARIUS_PM_NWG=$n PM_TRIALS=2 timeout -s KILL 90 python -u tools/pm_debug.py 2>&1 | grep -v amdgpu.ids | tee $out/pm_nwg$n.txt | cut -c1-300 | tail -12
done
