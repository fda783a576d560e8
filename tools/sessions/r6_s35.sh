#!/bin/bash
o=gpurun_out/r6s35; mkdir -p $o
( time python tools/vendor_step.py > $o/vendor.json 2> $o/vendor.err ) 2> $o/time.txt; tail -3 $o/time.txt; cat $o/vendor.json | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps(d.get('eval')), d.get('errors'))"
