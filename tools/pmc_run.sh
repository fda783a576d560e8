#!/bin/bash
# rocprofv3 counter passes over tools/pmc_kernels.py (GPU box).  Counters only with --kernel-trace (gpurun refuses
# --pmc together with the sys/hip/hsa trace domains).  Results: gpurun_out/pmc_<pass>.json
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $repo/gpurun_out/pmc_$name -o p -- python $repo/tools/pmc_kernels.py > $repo/gpurun_out/pmc_$name.log 2>&1
  db=$(find $repo/gpurun_out/pmc_$name -name "*.db" | head -1)
  python $repo/tools/pmc_summary.py $db $repo/gpurun_out/pmc_$name.json > $repo/gpurun_out/pmc_$name.txt 2>&1
  grep PMCMETA $repo/gpurun_out/pmc_$name.log | head -1 | sed 's/^PMCMETA //' > $repo/gpurun_out/pmc_meta.json
  rm -rf $repo/gpurun_out/pmc_$name
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run derived MfmaUtil VALUBusy
# the eval-mode embedding forward's kernels (matrix-pipe / LDS counters only)
run_embed() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $repo/gpurun_out/pmc_embed_$name -o p -- python $repo/tools/pmc_embed.py > $repo/gpurun_out/pmc_embed_$name.log 2>&1
  db=$(find $repo/gpurun_out/pmc_embed_$name -name "*.db" | head -1)
  python $repo/tools/pmc_summary.py $db $repo/gpurun_out/pmc_embed_$name.json > $repo/gpurun_out/pmc_embed_$name.txt 2>&1
  rm -rf $repo/gpurun_out/pmc_embed_$name
}
run_embed sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run_embed derived MfmaUtil VALUBusy
cd $repo
ls -la gpurun_out/pmc_*.json
