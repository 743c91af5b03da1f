import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:28]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]==sys.argv[2]: cnt[k]+=1
names=sys.argv[2:]
for k in sorted(agg, key=lambda k:-agg[k][names[0]])[:16]:
    a=agg[k]; n=max(cnt[k],1)
    print("%-28s n=%5d " % (k,n) + " ".join("%s=%.0f" % (c, a[c]/n) for c in names))
