#!/bin/bash
timeout 900 python -m pytest tests/test_research_gpu.py -m gpu -q 2>&1 | tail -3
