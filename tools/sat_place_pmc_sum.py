import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_sat_fast2" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(per)
for g in range(0, len(ids), 8):
    grp = ids[g:g + 8][3:]
    names = sorted(per[grp[0]])
    print(f"allocation {g // 8}: " + "  ".join(f"{n}={sum(per[i][n] for i in grp) / len(grp):.3e}" for n in names))
