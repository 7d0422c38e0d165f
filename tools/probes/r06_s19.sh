#!/bin/bash
O=gpurun_out/r06_s19; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "first_convolution_inside or operand_layout" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.txt
for f in 1 0; do timeout -k 5 120 python tools/profile_model.py landmark 512 stem_fuse=$f 2>/dev/null | grep -v amdgpu > $O/launches_fuse$f.txt; head -4 $O/launches_fuse$f.txt; tail -1 $O/launches_fuse$f.txt; done
