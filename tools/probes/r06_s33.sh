#!/bin/bash
set -o pipefail
cd /tmp 2>/dev/null; cd - >/dev/null
export TMPDIR=/tmp
O=gpurun_out/r06_s33; rm -rf $O; mkdir -p $O
for v in 0 1; do
  if [ $v = 1 ]; then export MI_NO_STEM_MFMA=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$v -- python3 tools/probes/u8_stem_probe.py > $O/t$v.log 2>&1; echo "rc $?"
  f=$(find $O/t$v -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]: print("%-70s calls %4s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
find $O -name "*.db" -delete 2>/dev/null
