import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter(); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    name=r["Kernel_Name"].split("(")[0].replace("void ","")
    acc[name][r["Counter_Name"]]+=float(r["Counter_Value"])
    key=(name,r["Dispatch_Id"])
    if key not in seen: seen.add(key); calls[name]+=1
for name,c in sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0)):
    wc=c.get("SQ_WAVE_CYCLES",0)
    if wc<=0 or calls[name]<20: continue
    print(f"{name[:44]:44s} n={calls[name]:4d} " + " ".join(f"{k.replace('SQ_','')}={v/wc:.3f}" for k,v in sorted(c.items()) if k!="SQ_WAVE_CYCLES"))
