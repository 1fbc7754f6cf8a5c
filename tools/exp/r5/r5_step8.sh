#!/bin/bash
timeout 900 python3 -m pytest tests/test_proj_ln_gpu.py -q -x 2>&1 | tail -4
timeout 400 python3 tools/proj_ln_bench.py 2>&1 | tail -36
