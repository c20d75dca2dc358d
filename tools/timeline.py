"""One-off: per-kernel timeline of the last encode step and of the last decode from a rocprofv3 --kernel-trace csv
(start offset, duration, queue): shows which kernels overlap and where the device waits.
usage: timeline.py <dir with *kernel_trace.csv> [first kernel of a step] [how many steps from the end]"""
import csv, glob, sys
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_init"
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == first]
if not starts:
    sys.exit("no kernel named " + first)
i0 = starts[-1]
t0 = rows[i0][0]
end = max(r[1] for r in rows[i0:i0 + 200] if r[0] - t0 < 5_000_000)
print("step from %s: %.3f ms" % (first, (end - t0) / 1e6))
for s, e, n, q in rows[i0:]:
    if s - t0 > 5_000_000:
        break
    print("%8.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
