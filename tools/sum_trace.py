import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
# forward(save=False) sweep = first nt steps: take launches between the first and the 2nd 'step3d' of variant save...
acc=collections.defaultdict(lambda:[0,0.0]); gaps=0.0; last=None; t0=None
for r in rows:
    n=r["Kernel_Name"]
    if "pml_kernel" not in n and "step3d" not in n: continue
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    key=n.split("(")[0][-60:]
    acc[key][0]+=1; acc[key][1]+=(e-s)/1e3
    if last is not None and s-last<50000: gaps+=max(0,(s-last))/1e3
    last=e
tot=sum(v[1] for v in acc.values())
for k,v in sorted(acc.items(),key=lambda kv:-kv[1][1]): print("%-62s n=%5d avg=%7.2f us total=%9.1f"%(k,v[0],v[1]/v[0],v[1]))
print("sum kernels %.1f us, sum small gaps %.1f us"%(tot,gaps))
