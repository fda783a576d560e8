#!/bin/bash
bash tools/ab.sh "CREID_TUNED_PLANS=centroids-reid_amd/tuned_plans_prev.json" "CREID_TUNED_PLANS=centroids-reid_amd/tuned_plans.json"
