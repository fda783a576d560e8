#!/bin/bash
python -m pytest tests/test_ctl_step_gpu.py -q -m gpu -k "reference_recording" -s 2>&1 | grep "^torch\.\|passed\|failed\|Error\|trajectory\|after 4 steps\|assert" | tail -30
