#!/bin/bash
python -m pytest tests/test_compute_gpu.py -m gpu -q -x 2>&1 | tail -8
