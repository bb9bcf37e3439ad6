set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p128t -o run -- python3 $R/bench.py --batch 128 --no-extras --no-cpu-baseline --steps 10 --warmup 4 > $R/gpurun_out/p128t.log 2>&1
cd $R
python3 tools/summarize_prof.py bygrid gpurun_out/p128t/run_kernel_trace.csv gpurun_out/r04_l_bygrid_b128.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/p128t/run_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: print the sequence of the final 200 kernels with durations and gaps
last=rows[-200:]
prev=None
out=[]
for r in last:
    st,en=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(st-prev)/1e3 if prev else 0
    out.append("%-60s grid %6d dur %7.1f gap %6.1f" % (r["Kernel_Name"].split("(")[0][-60:], int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"])), (en-st)/1e3, gap))
    prev=en
open("gpurun_out/r04_l_seq_b128.txt","w").write("\n".join(out))
PY
rm -f gpurun_out/p128t/run_kernel_trace.csv
