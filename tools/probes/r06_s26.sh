#!/bin/bash
set -o pipefail
cd /tmp 2>/dev/null; cd - >/dev/null
export TMPDIR=/tmp
O=gpurun_out/r06_s26; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -- python3 tools/profile_model.py back 256 > $O/sq.log 2>&1; echo "pmc rc $?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/sq2 -- python3 tools/profile_model.py back 256 > $O/sq2.log 2>&1; echo "pmc2 rc $?"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $O/sq3 -- python3 tools/profile_model.py back 256 > $O/sq3.log 2>&1; echo "pmc3 rc $?"
python3 - <<'PY'
import csv, glob, collections
for tag in ("sq", "sq2", "sq3"):
    fs = glob.glob('gpurun_out/r06_s26/%s/**/*counter_collection.csv' % tag, recursive=True)
    if not fs: print('no csv for', tag); continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'stem_mfma' not in r['Kernel_Name']: continue
        by[r['Counter_Name']].append(float(r['Counter_Value']))
    print("==", tag, {k: round(sorted(v)[len(v) // 2]) for k, v in by.items()})
PY
find $O -name "*.db" -delete 2>/dev/null
