"""Steady-state kernel summary from a rocprofv3 kernel_trace.csv: only the dispatches of the last `frac` of
the run (by dispatch order), grouped by a shortened kernel name.  Leaves out library warm-up (MIOpen's find
phase benchmarks every solver, the naive one included, on the first call of each convolution shape).
    python tools/ktrace_tail.py kernel_trace.csv [frac=0.4] [evals_in_tail]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
evals = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[int(len(rows) * (1 - frac)):]
agg = defaultdict(lambda: [0, 0.0])
for r in tail:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"<.*", "", n)[:60]
    a = agg[n]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
print("last %d dispatches: kernel time %.2f ms in a span of %.2f ms (per evaluation: %.2f / %.2f ms)" % (
    len(tail), tot / 1e3, span / 1e3, tot / 1e3 / evals, span / 1e3 / evals))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-60s %6d %9.2f ms %8.1f us %5.1f%%" % (n, a[0], a[1] / 1e3, a[1] / a[0], 100 * a[1] / tot))
