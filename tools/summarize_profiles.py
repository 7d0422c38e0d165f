"""Turns raw rocprofv3 output under gpurun_out/ into the small, committed summaries under profiles/.
usage: python tools/summarize_profiles.py <round-tag> <trace-dir> <fetch-dir> <write-dir>
PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from separate
passes, are reported in KiB, and on gfx950 FETCH_SIZE counts exactly half of a wide (16 B/lane) coalesced read stream,
so reads = 2 x FETCH_SIZE for the block / strip / chain kernels (their loads are 16 B/lane, plain or LDS-DMA); other kernels are
listed uncorrected."""
import collections, csv, glob, json, os, shutil, statistics, sys

tag, trace, fetch, write = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
ks = glob.glob(os.path.join(trace, "*", "*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(out, "%s_kernel_stats.csv" % tag))

def load(d):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0])):
        by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return by
f, w = load(fetch), load(write)
rows = []
for k in sorted(f, key=lambda k: -sum(f[k])):
    if not k.startswith("void mi::") and not k.startswith("mi::"):
        continue
    wide = any(t in k for t in ("block_kernel", "strip_kernel", "strip_pipe_kernel", "strip_pipe2_kernel", "chain_kernel"))
    fk, wk = statistics.mean(f[k]), statistics.mean(w.get(k, [0]))
    rows.append({"kernel": k, "dispatches": len(f[k]), "FETCH_SIZE_KiB_avg": round(fk, 1), "WRITE_SIZE_KiB_avg": round(wk, 1),
                 "fetch_correction": 2 if wide else 1, "hbm_bytes_per_launch": round(((2 if wide else 1) * fk + wk) * 1024)})
with open(os.path.join(out, "%s_pmc_by_kernel.csv" % tag), "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    wr.writeheader()
    wr.writerows(rows)
dom = max(csv.DictReader(open(ks)), key=lambda r: float(r["TotalDurationNs"]))
name = dom["Name"]
pm = next(r for r in rows if r["kernel"] == name)
import re
def label(name):
    # rocprofv3 symbol -> the label mi_model_profile / bench.py use: template arguments as integers, SLOW / CPT dropped for block kernels
    m = re.search(r"(\w+)<([^>]*)>", name.replace("(anonymous namespace)::", ""))
    if not m: return name
    args = [{"true": "1", "false": "0"}.get(a.strip(), a.strip()) for a in m.group(2).split(",")]
    if m.group(1) == "block_kernel": args = args[:5]
    return "%s<%s>" % (m.group(1), ",".join(args))
label = label(name)
json.dump({"round": tag, "workload": "back256_b256", "kernel": label, "rocprof_name": name, "calls": int(dom["Calls"]),
           "avg_ns": float(dom["AverageNs"]), "hbm_bytes_per_launch": pm["hbm_bytes_per_launch"],
           "note": "reads = 2 x FETCH_SIZE (gfx950 16B/lane stream correction) + WRITE_SIZE, KiB -> bytes"},
          open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(open(os.path.join(out, "pmc_summary.json")).read())
# per-launch event profiles of every model (tools/profile_model.py) and the secondary configs, when collect_profiles.sh produced them
base = os.path.dirname(os.path.normpath(trace))
for fpath in glob.glob(os.path.join(base, "launches_*.txt")):
    shutil.copy(fpath, os.path.join(out, "%s_%s" % (tag, os.path.basename(fpath))))
if os.path.exists(os.path.join(base, "configs.log")):
    with open(os.path.join(out, "configs_%s.jsonl" % tag), "w") as fh:
        fh.writelines(l for l in open(os.path.join(base, "configs.log")) if l.startswith("{"))
if os.path.exists(os.path.join(base, "bench.json")):
    with open(os.path.join(out, "bench_%s_n1.json" % tag), "w") as fh:
        fh.write(open(os.path.join(base, "bench.json")).read().strip().splitlines()[-1] + "\n")
