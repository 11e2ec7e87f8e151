"""Parse a rocprofv3 --kernel-trace CSV of a multi-stream run: per stream (queue) the kernel time and the idle gaps between
consecutive kernels, and how much of the wall span had 0 / 1 / 2+ kernels in flight.  python tools/trace_overlap.py DIR [tail_fraction]"""
import csv, glob, sys, collections
d = sys.argv[1]; tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"][:40]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + (t1 - t0) * (1 - tail)
rows = [r for r in rows if r[0] >= cut]
span = rows[-1][1] - rows[0][0]
byq = collections.defaultdict(list)
for r in rows: byq[r[2]].append(r)
print(f"span {span/1e3:.0f} us, {len(rows)} kernels, {len(byq)} queues")
for q, rs in byq.items():
    busy = sum(e - s for s, e, _, _ in rs)
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    names = collections.Counter(n for _, _, _, n in rs)
    per = collections.defaultdict(list)
    for s, e, _, n in rs: per[n].append(e - s)
    print(f"queue {q}: {len(rs)} kernels, busy {busy/span:.2f} of span, mean gap {sum(gaps)/max(1,len(gaps))/1e3:.1f} us")
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:5]:
        print(f"    {n:42s} n={len(v):5d} mean {sum(v)/len(v)/1e3:7.1f} us")
ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, dlt in ev:
    hist[min(depth, 3)] += t - last; last = t; depth += dlt
print("kernels in flight (fraction of span):", {k: round(v / span, 3) for k, v in sorted(hist.items())})
