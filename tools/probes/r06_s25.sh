#!/bin/bash
# round 6: effective shader clock of the 5x5 stem's two forms (and of the MFMA form without its loads): GRBM_GUI_ACTIVE / duration
set -o pipefail
cd /tmp 2>/dev/null; cd - >/dev/null
export TMPDIR=/tmp
O=gpurun_out/r06_s25; rm -rf $O; mkdir -p $O
for v in "0 1" "0 0"; do set -- $v   # (the third variant of the session, the MFMA form without its loads, was a development build)
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/gui_$1_$2 -- python3 tools/profile_model.py back 256 stem_mfma=$2 > $O/gui_$1_$2.log 2>&1; echo "pmc abl $1 mfma $2 rc $?"
done
python3 - <<'PY'
import csv, glob, collections
for tag in ("0_1", "0_0"):
    fs = glob.glob('gpurun_out/r06_s25/gui_%s/**/*counter_collection.csv' % tag, recursive=True)
    if not fs: print('no csv for', tag); continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
        dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        name = r['Kernel_Name'].split('(')[0].replace('void mi::(anonymous namespace)::', '')[:48]
        by[name].append((float(r['Counter_Value']), dur))
    print("== abl_mfma", tag)
    for k, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1]))[:4]:
        v = v[len(v) // 4:]; v.sort(key=lambda t: t[1]); cyc, dur = v[len(v) // 2]
        print('%-50s n %3d  %8.1f us  %10.0f cyc  -> %6.2f GHz (8 XCDs, ~9.9 us of set-up included)' % (k, len(v), dur / 1e3, cyc, cyc / 8 / (dur / 1e3 + 9.9) / 1e3))
PY
find $O -name "*.db" -delete 2>/dev/null
