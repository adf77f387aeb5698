"""Sum rocprofv3 --pmc CSVs per kernel: mean counter value per dispatch (tools only).  usage: pmc_sum.py DIR [kernel-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if flt and flt not in k:
            continue
        a = acc[k.split("(")[0][:60]][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print(f"   {c:45s} {s / n:16.1f}   (x{n})")
