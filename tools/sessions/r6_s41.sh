#!/bin/bash
# fp32 step: N-tile rule of the fp32 implicit-GEMM kernel (128 columns unless fewer than MIN_WGS workgroups) re-measured with the new k-loop
bash tools/ab.sh "CREID_BENCH_DTYPE=f32 CREID_IGEMM_BN128_MIN_WGS=384" "CREID_BENCH_DTYPE=f32 CREID_IGEMM_BN128_MIN_WGS=256" "CREID_BENCH_DTYPE=f32 CREID_IGEMM_BN128_MIN_WGS=128" "CREID_BENCH_DTYPE=f32 CREID_IGEMM_BN128_MIN_WGS=1024"
