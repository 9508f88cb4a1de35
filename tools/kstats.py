"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: one line per kernel."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = sys.argv[2] if len(sys.argv) > 2 else None
tot = sum(float(r["TotalDurationNs"]) for r in rows if not (skip and skip in r["Name"]))
print("%-72s %7s %10s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    if skip and skip in r["Name"]:
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"\((sid_attn::|PackParams|void const).*", "", n)[:72]
    print("%-72s %7s %10.3f %9.1f %9.1f %9.1f %6.2f" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                      float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.3f ms" % (tot / 1e6))
