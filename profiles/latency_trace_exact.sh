export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06_lat_exact -o lat -- python3 $R/profiles/latency_probe.py exact 200 2000 > $R/gpurun_out/r06_lat_exact.log 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/r06_lat_exact/lat_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//4:]          # past the warm-up
d=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-60:]; d[k][0]+=1; d[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
n=max(c for c,_ in d.values())
for k,(c,t) in sorted(d.items(), key=lambda x:-x[1][1])[:12]: print("%-62s %5d launches, %8.1f us per launch, %6.1f us per call"%(k,c,t/c,t/n))
print("GPU time per call: %.1f us" % (sum(t for _,t in d.values())/n))
PY
grep "median" $R/gpurun_out/r06_lat_exact.log
rm -f $R/gpurun_out/r06_lat_exact/lat_kernel_trace.csv
