#!/bin/bash
# flakiness check: the whole GPU suite twice, in random-ish order the second time (by file, reversed)
o=gpurun_out/r6s39; mkdir -p $o
python -m pytest tests -m gpu -q -x > $o/run1.log 2>&1; echo "run1 rc $?"; tail -2 $o/run1.log
python -m pytest $(ls tests/test_*gpu*.py tests/test_transforms.py 2>/dev/null | sort -r) -m gpu -q -x -p no:cacheprovider > $o/run2.log 2>&1; echo "run2 rc $?"; tail -2 $o/run2.log
