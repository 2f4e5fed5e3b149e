"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean of each counter per dispatch."""
import csv, glob, sys, collections, os
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")[:90]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
