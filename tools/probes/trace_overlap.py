import csv, glob, sys
ev=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Queue_Id","?"), r.get("Stream_Id","?")))
ev.sort()
print(len(ev), "kernels; queues", sorted(set(e[3] for e in ev)), "streams", len(set(e[4] for e in ev)))
# overlap statistics per third of the run
n=len(ev)
for part in range(6):
    seg=ev[part*n//6:(part+1)*n//6]
    ov=0; tot=0
    for a,b in zip(seg,seg[1:]):
        tot+=1
        if b[0] < a[1]-1000: ov+=1
    busy=sum(e[1]-e[0] for e in seg); span=seg[-1][1]-seg[0][0]
    print("part",part,"overlapping successive pairs",ov,"of",tot,"sum of durations / span = %.3f"%(busy/span), "queues", sorted(set(e[3] for e in seg)))

if len(sys.argv) > 2:
    k0 = int(sys.argv[2])
    t0 = ev[k0][0]
    for e in ev[k0:k0+40]:
        print("%9.1f %9.1f q%s %s" % ((e[0]-t0)/1e3, (e[1]-t0)/1e3, e[3], e[2].replace("void mi::(anonymous namespace)::","").replace("mi::(anonymous namespace)::","")))

# ---- busy / overlap fractions over the steady part of the run (argv[3] = "busy")
if len(sys.argv) > 3 and sys.argv[3] == "busy":
    seg = [e for e in ev if "mi::" in e[2] or "mi::" in e[2]]
    seg = seg[len(seg) // 3: 2 * len(seg) // 3]
    pts = []
    for s_, e_, *_ in seg:
        pts.append((s_, 1)); pts.append((e_, -1))
    pts.sort()
    depth = 0; last = pts[0][0]; acc = {}
    for t, d in pts:
        acc[depth] = acc.get(depth, 0) + (t - last)
        last = t; depth += d
    span = pts[-1][0] - pts[0][0]
    print("middle third of the run: %d kernels over %.2f ms; time with 0 / 1 / 2 / 3+ kernels running: %s" % (
        len(seg), span / 1e6, " / ".join("%.3f" % (acc.get(k, 0) / span) for k in (0, 1, 2)) + " / %.3f" % (sum(v for k, v in acc.items() if k >= 3) / span)))
