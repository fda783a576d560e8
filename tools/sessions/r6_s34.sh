#!/bin/bash
# Round-6 session 34: fp32 implicit-GEMM kernel with the two-phase k-loop: fp32 parity tests, then the fp32 step A/B against the round-5 library
o=gpurun_out/r6s34; mkdir -p $o
python -m pytest tests/test_backbone_gpu.py tests/test_ctl_step_gpu.py tests/test_parity_full_size_gpu.py tests/test_round2_gpu.py tests/test_eval_fold_gpu.py -q -m gpu -x > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $o/pytest.log
bash tools/ab.sh "CREID_BENCH_DTYPE=f32 CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_prev.so" "CREID_BENCH_DTYPE=f32" > $o/ab.txt 2>&1; cat $o/ab.txt
