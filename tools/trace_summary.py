"""Per-kernel totals of a per-dispatch step trace written by tools/trace_step.sh:  python tools/trace_summary.py <csv> [--all]"""
import sys
rows = []
for ln in open(sys.argv[1]).read().splitlines()[1:]:
    p = ln.rsplit(',', 5)
    rows.append((p[0], p[1], float(p[3]), float(p[4]), float(p[5])))
tot = sum(r[3] for r in rows)
agg = {}
for k, g, s, d, gap in rows:
    a = agg.setdefault(k[:80], [0, 0.0]); a[0] += 1; a[1] += d
print("total us %.1f over %d dispatches" % (tot, len(rows)))
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-82s %4d %9.1f %5.1f%%  %7.1f" % (k, n, d, 100 * d / tot, d / n))
if "--all" in sys.argv:
    for i, (k, g, s, d, gap) in enumerate(rows):
        print("%3d %-70s %-16s %9.1f %7.1f %5.1f" % (i, k[:70], g, s, d, gap))
