"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            if "render_kernel" not in k:
                continue
            acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
