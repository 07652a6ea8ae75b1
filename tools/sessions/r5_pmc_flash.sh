#!/bin/bash
# matrix-pipe counters of the flash launches (tools/bench_flash.py: the decoder trio at the bench shape, fp16 records), folded column tail on / off
tag=${1:-pmcf}
ulimit -c 0
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for v in 1 0; do
  MARIUS_FLASH_TAIL4=$v bash tools/pmc_flash.sh ${tag}_tail$v SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE > gpurun_out/$tag/counters_tail$v.txt 2>&1
  grep "flash_kernel<7" gpurun_out/$tag/counters_tail$v.txt | cut -c1-400
  MARIUS_FLASH_TAIL4=$v timeout 200 python tools/bench_flash.py 2>/dev/null | head -2
done
