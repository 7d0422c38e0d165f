#!/bin/bash
# round 6, GPU session 4: effective shader clock per kernel of the real steps: GRBM_GUI_ACTIVE (sclk cycles the GPU was busy during the dispatch) / duration
set -o pipefail
cd /tmp 2>/dev/null; cd - >/dev/null
export TMPDIR=/tmp
O=gpurun_out/r06_s4; rm -rf $O; mkdir -p $O
for c in 2 5 3; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/gui_c$c -- python3 bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-latency --no-host-feed --no-secondary --no-event-profile --single-window --in-flight 1 > $O/gui_c$c.log 2>&1; echo "pmc $c rc $?"
done
python3 - <<'PY'
import csv, glob, collections, os
for c in (2, 5, 3):
    fs = glob.glob('gpurun_out/r06_s4/gui_c%d/**/*counter_collection.csv' % c, recursive=True)
    if not fs: print('no csv for', c); continue
    rows = list(csv.DictReader(open(fs[0])))
    by = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
        dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        name = r['Kernel_Name'].split('(')[0].replace('void mi::(anonymous namespace)::', '')[:60]
        by[(name, r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''))].append((float(r['Counter_Value']), dur))
    out = open('gpurun_out/r06_s4/clock_by_kernel_c%d.txt' % c, 'w')
    print('# config %d: per kernel (name, grid): dispatches, median duration us, GRBM_GUI_ACTIVE per XCC? / duration -> MHz' % c, file=out)
    for k, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        v = v[len(v) // 4:]  # steady part
        v.sort(key=lambda t: t[1])
        cyc, dur = v[len(v) // 2]
        print('%-62s grid %-9s n %3d  %8.1f us  %10.0f cyc  -> %7.1f MHz' % (k[0], k[1], len(v), dur / 1e3, cyc, cyc / dur * 1e3), file=out)
    out.close()
    print(open(out.name).read())
PY
find $O -name "*.db" -delete 2>/dev/null; du -sh $O
