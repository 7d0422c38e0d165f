#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_pipeline_on_man_jpg" 2>&1 | tail -12
