#!/bin/bash
# Round-6 session 21: matrix-pipe busy and wait counters of the streamed count kernel (new k-loop)
repo=$(pwd); o=$repo/gpurun_out/r6s21; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $o/pmc_$name -o p -- python $repo/tools/debug/count_probe.py > $o/pmc_$name.log 2>&1
  db=$(find $o/pmc_$name -name "*.db" | head -1)
  python $repo/tools/pmc_summary.py $db $o/pmc_$name.json > $o/pmc_$name.txt 2>&1
  rm -rf $o/pmc_$name
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32
cd $repo
python - <<'PY'
import json
for n in ("sq", "wait"):
    d = json.load(open(f"gpurun_out/r6s21/pmc_{n}.json"))
    for k, v in d.items():
        if k.startswith(("sqdist", "_")) is False: continue
        if k.startswith("_"): continue
        print(n, k[:40], {c: (x["sum"], x["dispatches"]) for c, x in v.items()})
PY
