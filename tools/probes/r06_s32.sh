#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "round6_launches_in_chunks" 2>&1 | tail -15
