import csv, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if k.startswith("rt::"): print(k, dict(v))
