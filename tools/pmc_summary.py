"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:40]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
