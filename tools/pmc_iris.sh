# SQ counters of the iris stage programs (separate rocprofv3 --pmc passes, 4 counters each); summarise with the snippet in DESIGN.md section 8
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_IFETCH SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc/d -- python3 tools/profile_model.py iris 1024 > gpurun_out/pmc/d.log 2>&1; echo d $?
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc/e -- python3 tools/profile_model.py iris 1024 > gpurun_out/pmc/e.log 2>&1; echo e $?
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc/f -- python3 tools/profile_model.py iris 1024 > gpurun_out/pmc/f.log 2>&1; echo f $?
