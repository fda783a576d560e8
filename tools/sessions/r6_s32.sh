#!/bin/bash
# Round-6 session 32: warp-specialised igemm kernel, consumer loop with the next k-tile's first fragments read under the last slice's MFMAs
o=gpurun_out/r6s32; mkdir -p $o
python -m pytest tests/test_conv_gpu.py tests/test_conv_pipe_gpu.py tests/test_backbone_gpu.py tests/test_bench_path_gpu.py -q -m gpu -x > $o/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $o/pytest.log
bash tools/ab.sh "CREID_LIB_PATH=$PWD/centroids-reid_amd/lib/libcreid_hip_prev.so" "CREID_X=new" > $o/ab.txt 2>&1; cat $o/ab.txt
