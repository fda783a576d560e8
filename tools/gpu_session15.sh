#!/bin/bash
bash tools/ab.sh "CREID_LIB_PATH=centroids-reid_amd/lib/libcreid_hip.so" "CREID_LIB_PATH=centroids-reid_amd/lib/alt/libcreid_occ3.so" "CREID_LIB_PATH=centroids-reid_amd/lib/alt/libcreid_occ3.so CREID_TUNED_PLANS=0" "CREID_LIB_PATH=centroids-reid_amd/lib/libcreid_hip.so CREID_TUNED_PLANS=0"
