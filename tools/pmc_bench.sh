# SQ counters of the BASELINE workload's kernels (separate passes, 4 counters each): where the waves' cycles go
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcb
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcb/p$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/pmcb/p$i.log 2>&1; echo pass $i $?
done
