"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the small, committed summaries under
profiles/.   usage: python tools/summarize_profiles.py r02
PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from separate passes, are
reported in KiB, and on gfx950 FETCH_SIZE counts exactly half of a wide (16 B/lane) coalesced read stream, so reads = 2 x FETCH_SIZE
for the block / strip / chain kernels (their loads are 16 B/lane, plain or LDS-DMA); other kernels are listed uncorrected.
Other kernels (stem, generic, post-processing) are listed uncorrected: their access widths are not calibrated.
pmc_summary.json is stamped with a hash of the kernel sources: bench.py reports `traffic` only while that hash still matches."""
import collections, csv, glob, json, os, re, shutil, statistics, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
base = os.path.join(root, "gpurun_out", "prof_" + tag)
# second argument: where the summaries go (tools/collect_profiles.sh summarises ON the GPU box into gpurun_out/prof_<tag>/summary, so
# that the bench lines it takes afterwards read the PMC summary of the same collection; copy that directory into profiles/ afterwards)
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(base, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None   # gpurun merges into gpurun_out/: files of an earlier collection may still be there


def label(name):
    # rocprofv3 symbol -> the label mi_model_profile / bench.py use: template arguments as integers, SLOW / CPT dropped for block kernels
    name = name.replace("(anonymous namespace)::", "")
    # the operand-layout kernels carry a geometry struct as template argument: mdblock_kernel<mi::MD<8, 2, 2, 2, 3, false, 4, 1, false, false>, true>
    m1 = re.search(r"ms2_kernel<(?:mi::)?MS2<([^>]*)>", name)
    if m1:   # MS2<CK, MT, CO, NWV, WO, ...> -> the engine's label <C / 4, output tiles, pixel tiles>
        a = [x.strip() for x in m1.group(1).split(",")]
        return "ms2_kernel<%s,%s,%s>" % (a[0], a[1], a[3])
    m0 = re.search(r"(mdblock_kernel|mbneck_kernel|mwalk_kernel)<(?:mi::)?M[DW]<([^>]*)>", name)
    if m0:
        a = [x.strip() for x in m0.group(2).split(",")]
        if m0.group(1) == "mwalk_kernel":
            return "mwalk_kernel<%s>" % ",".join(a[:3])
        if m0.group(1) == "mdblock_kernel" and len(a) >= 11 and a[9] == "true" and a[10] == "true":
            return "mdblock_kernel<stem+pair>"   # (round 6: the face mesh's first convolution inside the pair's launch)
        if m0.group(1) == "mdblock_kernel" and len(a) >= 10 and a[9] == "true":
            return "mdblock_kernel<pair>"
        return m0.group(1)
    m = re.search(r"(\w+)<([^>]*)>", name)
    if not m:
        m2 = re.search(r"(\w+)\(", name)
        return m2.group(1) if m2 else name
    args = [{"true": "1", "false": "0"}.get(a.strip(), a.strip()) for a in m.group(2).split(",")]
    if m.group(1) == "block_kernel":
        args = args[:5]
    if m.group(1) == "chain_kernel":
        args = args[:1]   # the label carries the tile count only (not the unit-split flag)
    if m.group(1) in ("bneck_kernel", "dblock_kernel", "tail_kernel"):
        return m.group(1)
    if m.group(1) in ("stem_conv_kernel", "stem_mfma_kernel"):
        return m.group(1)
    return "%s<%s>" % (m.group(1), ",".join(args))


WORKLOADS = {1: "short128_b256", 2: "back256_b256", 3: "landmark192_b512", 5: "pipeline192_b128"}
WIDE = ("tail_kernel", "block_kernel", "strip_kernel", "strip_pipe_kernel", "strip_pipe2_kernel", "strip_pipe2m_kernel", "chain_kernel", "bneck_kernel", "dblock_kernel", "mstrip_kernel", "mdblock_kernel", "mbneck_kernel", "mwalk_kernel", "ms2_kernel", "xc_kernel")

# ---- kernel stats per config
for c in WORKLOADS:
    ks = one("trace_c%d/**/*_kernel_stats.csv" % c)
    if ks:
        shutil.copy(ks, os.path.join(out, "%s_kernel_stats_config%d.csv" % (tag, c)))
    ks2 = one("trace2_c%d/**/*_kernel_stats.csv" % c)   # the same run with two batches in flight (the mode bench.py times by default)
    if ks2:
        shutil.copy(ks2, os.path.join(out, "%s_kernel_stats_config%d_in_flight2.csv" % (tag, c)))


STEADY = {}   # per config: kernel label -> steady-state nanoseconds per launch

# ---- the kernel-trace rows the per-symbol averages of --stats come from, per (kernel, grid size): a symbol that runs on two shapes (the
# 128^2 and 64^2 row pipelines) is two lines here
for c in WORKLOADS:
    kt = one("trace_c%d/**/*_kernel_trace.csv" % c)
    if not kt:
        continue
    # the shapes a symbol runs on, in launch order within one step: the per-launch event list of the same model (tools/profile_model.py);
    # the k-th dispatch of a symbol is its (k mod launches-per-step)-th launch of the step (two shapes can share a grid size: the 128^2
    # and 64^2 row pipelines are both 256 workgroups)
    order = collections.defaultdict(list)
    lf = os.path.join(base, "launches_%s.txt" % {1: "front", 2: "back", 3: "landmark"}.get(c, "none"))
    if os.path.exists(lf):
        for line in open(lf):
            parts = line.split()
            if len(parts) >= 4 and parts[1] == "ms":
                order[parts[2]].append(parts[3])
    rows_kt = sorted((r for r in csv.DictReader(open(kt)) if "mi::" in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
    seen = collections.Counter()
    per = collections.defaultdict(list)
    for r in rows_kt:
        lab = label(r["Kernel_Name"])
        shapes = order.get(lab, [])
        shape = shapes[seen[lab] % len(shapes)] if shapes else ""
        seen[lab] += 1
        grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        wg = "x".join(str(r.get(k, "")) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
        per[(lab, shape, grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if c == 2:  # the rows themselves for the headline workload (a few hundred lines): what the averages above are averages of
        t0 = min(int(r["Start_Timestamp"]) for r in rows_kt)
        with open(os.path.join(out, "%s_kernel_trace_rows_config%d.csv" % (tag, c)), "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(["dispatch_id", "kernel", "grid_threads", "workgroup_threads", "start_ns_from_first", "duration_ns"])
            for r in rows_kt:
                short = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                wr.writerow([r["Dispatch_Id"], short, r["Grid_Size_X"], r["Workgroup_Size_X"], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    with open(os.path.join(out, "%s_kernel_trace_by_shape_config%d.csv" % (tag, c)), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "shape", "grid_threads", "workgroup", "dispatches", "avg_ns", "median_ns", "min_ns", "max_ns", "steady_median_ns"])
        for (lab, shape, grid, wg), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            # steady state = without the dispatches of the first two steps (the chip's clocks ramp up over them)
            wr.writerow([lab, shape, grid, wg, len(v), round(statistics.mean(v)), round(statistics.median(v)), min(v), max(v), round(statistics.median(v[2:] or v))])
    # per SYMBOL (what `rocprofv3 --stats` and the bench line's `roofline` report): launches per step, the --stats average, and the
    # steady-state figure = mean over the symbol's launches of a step of each launch's median without its first two dispatches
    by_sym = collections.defaultdict(list)
    for (lab, shape, grid, wg), v in per.items():
        by_sym[lab].append(v)
    STEADY[c] = {}
    with open(os.path.join(out, "%s_kernel_steady_config%d.csv" % (tag, c)), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "launches_per_step", "dispatches", "avg_ns_all_dispatches", "steady_ns_per_launch"])
        for lab, vs in sorted(by_sym.items(), key=lambda kv: -sum(sum(v) for v in kv[1])):
            # (order[lab] lists the symbol's launches of one step; a shape launched twice per step owns two of them)
            lps = max(len(order.get(lab, [])), len(vs))
            n_all = sum(len(v) for v in vs)
            steady = sum(statistics.median(v[2 * max(1, round(len(v) * lps / n_all)):] or v) * len(v) for v in vs) / n_all
            STEADY[c][lab] = steady
            wr.writerow([lab, lps, n_all, round(sum(sum(v) for v in vs) / n_all), round(steady)])


def load(d):
    by = collections.defaultdict(list)
    f = one(d + "/**/*_counter_collection.csv")
    if not f:
        return by
    for r in csv.DictReader(open(f)):
        by[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return by


import bench
entries, rows = [], []
for c, workload in WORKLOADS.items():
    # ---- HBM traffic of this config's kernels
    f, w = load("fetch_c%d" % c), load("write_c%d" % c)
    crow = []
    for (k, _c) in sorted(f, key=lambda kc: -sum(f[kc])):
        if "mi::" not in k:
            continue
        wide = any(t in k for t in WIDE)
        fk, wk = statistics.mean(f[(k, "FETCH_SIZE")]), statistics.mean(w.get((k, "WRITE_SIZE"), [0]))
        crow.append({"config": c, "kernel": k, "label": label(k), "dispatches": len(f[(k, "FETCH_SIZE")]), "FETCH_SIZE_KiB_avg": round(fk, 1), "WRITE_SIZE_KiB_avg": round(wk, 1),
                     "fetch_correction": 2 if wide else 1, "hbm_bytes_per_launch": round(((2 if wide else 1) * fk + wk) * 1024)})
    rows += crow
    ksc = os.path.join(out, "%s_kernel_stats_config%d.csv" % (tag, c))
    avg = {}
    if os.path.exists(ksc):
        tot = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(ksc)):
            t = tot[label(r["Name"])]
            t[0] += float(r["TotalDurationNs"]); t[1] += int(r["Calls"])
        avg = {k: v[0] / max(v[1], 1) for k, v in tot.items()}
    ksc2 = os.path.join(out, "%s_kernel_stats_config%d_in_flight2.csv" % (tag, c))
    avg2 = {}
    if os.path.exists(ksc2):
        tot = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(ksc2)):
            t = tot[label(r["Name"])]
            t[0] += float(r["TotalDurationNs"]); t[1] += int(r["Calls"])
        avg2 = {k: v[0] / max(v[1], 1) for k, v in tot.items()}
    by_label = collections.defaultdict(list)
    for r in crow:
        by_label[r["label"]].append(r)
    entries += [{"round": tag, "workload": workload, "kernel": lab, "source_hash": bench.kernel_source_hash(), "rocprof_avg_ns": avg.get(lab),
                 "rocprof_steady_ns": STEADY.get(c, {}).get(lab), "rocprof_in_flight2_avg_ns": avg2.get(lab),
                 "hbm_bytes_per_launch": round(sum(r["hbm_bytes_per_launch"] * r["dispatches"] for r in rs) / sum(r["dispatches"] for r in rs)),
                 "note": "reads = %d x FETCH_SIZE (gfx950: FETCH_SIZE counts half of a 16 B/lane stream) + WRITE_SIZE, KiB -> bytes; separate --pmc passes" % rs[0]["fetch_correction"]}
                for lab, rs in by_label.items()]
    # ---- SQ counters of this config (where a wave's cycles go)
    sq = {}
    for i in range(1, 4):
        for (k, cn), v in load("sq%d_c%d" % (i, c)).items():
            sq.setdefault(k, {})[cn] = statistics.mean(v)
    if sq:
        with open(os.path.join(out, "%s_sq_counters_config%d.txt" % (tag, c)), "w") as fh:
            fh.write("# rocprofv3 --pmc passes of tools/collect_profiles.sh on bench.py --config %d (%s): per kernel, averages per dispatch;\n"
                     "# cyc/wave = 4 x SQ_WAVE_CYCLES / SQ_WAVES; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier); stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls);\n"
                     "# valu / salu / lds = SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES per SIMD: see DESIGN.md); instruction counts per wave\n" % (c, workload))
            for k, cc in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
                if "mi::" not in k or not cc.get("SQ_WAVES"):
                    continue
                wv, wc = cc["SQ_WAVES"], max(cc.get("SQ_WAVE_CYCLES", 0), 1)
                g = lambda n: cc.get(n, 0)
                fh.write("%-34s waves %6d cyc/wave %8.0f wait %.2f stall %.2f (lds %.2f) valu %.2f salu %.2f | VALU %.0f MFMA %.0f SALU %.0f LDS %.0f SMEM %.0f VMEM %.0f | mfma-busy cyc/wave %.0f\n" % (
                    label(k), wv, 4 * wc / wv, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_SCA") / wc,
                    g("SQ_INSTS_VALU") / wv, g("SQ_INSTS_MFMA") / wv, g("SQ_INSTS_SALU") / wv, g("SQ_INSTS_LDS") / wv, g("SQ_INSTS_SMEM") / wv,
                    (g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")) / wv, g("SQ_VALU_MFMA_BUSY_CYCLES") / wv))
if rows:
    with open(os.path.join(out, "%s_pmc_by_kernel.csv" % tag), "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows)
    json.dump({"entries": entries}, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)

# ---- bench lines, per-launch event lists, secondary configs
for c in WORKLOADS:
    p = os.path.join(base, "bench_c%d.json" % c)
    dst = os.path.join(out, "bench_%s_config%d_n1.json" % (tag, c))
    if os.path.exists(p) and open(p).read().strip():
        line = open(p).read().strip().splitlines()[-1]
        with open(dst, "w") as fh:
            fh.write(line + "\n")
for fpath in glob.glob(os.path.join(base, "launches_*.txt")):
    shutil.copy(fpath, os.path.join(out, "%s_%s" % (tag, os.path.basename(fpath))))
if os.path.exists(os.path.join(base, "configs.log")):
    with open(os.path.join(out, "configs_%s.jsonl" % tag), "w") as fh:
        fh.writelines(l for l in open(os.path.join(base, "configs.log")) if l.startswith("{"))
print("wrote", sorted(f for f in os.listdir(out) if tag in f or f == "pmc_summary.json"))
