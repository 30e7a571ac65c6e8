#!/bin/bash
timeout 900 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -8 | cut -c1-400
