#!/bin/bash
python -m pytest tests/test_backbone_gpu.py -q -m gpu -k "autocast" -s 2>&1 | grep -v "^$" | tail -30
