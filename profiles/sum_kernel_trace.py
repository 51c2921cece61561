import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r["Kernel_Name"][:60]; d[k][0]+=1; d[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for k,(c,t) in sorted(d.items(), key=lambda x:-x[1][1])[:12]: print("%-62s %5d launches %10.2f ms  avg %8.3f"%(k,c,t,t/c))
