#!/bin/bash
# Round-6 final GPU session: everything under profiles/r06_* from one box.   gpurun --timeout 3000 -- bash tools/gpu_r6_final.sh
export CREID_ROUND=r06
o=gpurun_out/r6f; mkdir -p $o
repo=$(pwd)
python -m pytest tests -m gpu -q --maxfail=25 --durations=8 > $o/pytest.log 2>&1
echo "pytest rc $?" >> $o/pytest.log; tail -14 $o/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; tail -2 $o/smoke.log
( time python bench.py > $o/bench.json 2> $o/bench.err ) 2> $o/bench.time; echo "bench rc $?"; tail -3 $o/bench.time
# what each round-6 schedule change is worth on THIS box (same process environment, two interleaved repeats)
bash tools/ab.sh "CREID_HEADS_ONE_CALL=0 CREID_DS_REDUCE2=0 CREID_C3_AXF= CREID_HEADS_BNRED=0" "CREID_DS_REDUCE2=0 CREID_C3_AXF= CREID_HEADS_BNRED=0" \
  "CREID_C3_AXF= CREID_HEADS_BNRED=0" "CREID_C3_AXF=" "CREID_X=default" "CREID_WPREP_SIDE=1" > $o/switches_ab.txt 2>&1; cat $o/switches_ab.txt
bash tools/prof_train.sh r6f > $o/train_step_anatomy.md 2>&1; head -22 $o/train_step_anatomy.md
db=$(find gpurun_out/prof_r6f -name "*.db" | head -1)
python tools/train_layers.py $db > $o/train_layers.md 2>&1; tail -16 $o/train_layers.md
python tools/step_sequence.py $db > $o/step_sequence.txt 2>&1
cp gpurun_out/prof_r6f.md $o/train_kernel_stats.md 2>/dev/null
bash tools/prof_embed.sh r6fe > $o/embed_anatomy.md 2>&1; tail -4 $o/embed_anatomy.md
(cd /tmp && export TMPDIR=/tmp && CREID_BENCH_NO_PMC=1 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_r6fev -o ev -- \
   python $repo/bench.py --workload eval --steps 5 --warmup 2 --no-cpu-baseline > $repo/gpurun_out/prof_r6fev.log 2>&1)
python tools/prof_summary.py $(find gpurun_out/prof_r6fev -name "*.db" | head -1) $o/eval_kernel_stats.md > /dev/null; head -14 $o/eval_kernel_stats.md
# the exact-f32 parity mode of the step: where its time goes
(cd /tmp && export TMPDIR=/tmp && CREID_BENCH_DTYPE=f32 CREID_BENCH_NO_EVAL=1 CREID_BENCH_NO_INSITU=1 CREID_BENCH_NO_PMC=1 rocprofv3 --kernel-trace --stats -d $repo/gpurun_out/prof_r6f32 -o t -- \
   python $repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $repo/gpurun_out/prof_r6f32.log 2>&1)
python tools/step_anatomy.py $(find gpurun_out/prof_r6f32 -name "*.db" | head -1) > $o/fp32_step_anatomy.md 2>&1; head -16 $o/fp32_step_anatomy.md
rm -rf gpurun_out/prof_r6f gpurun_out/prof_r6fe gpurun_out/prof_r6fev gpurun_out/prof_r6f32
tools/probes/anyorder_probe > $o/anyorder_probe.txt 2>&1; tail -12 $o/anyorder_probe.txt
python tools/vendor_step.py > $o/vendor_yardstick.json 2> $o/vendor.err; cat $o/vendor_yardstick.json
bash tools/pmc_run.sh > $o/pmc.log 2>&1; tail -6 $o/pmc.log
python tools/pmc_report.py > $o/pmc_report.log 2>&1; tail -5 $o/pmc_report.log
cp profiles/r06_pmc_traffic.json profiles/r06_pmc_summary.md $o/ 2>/dev/null
