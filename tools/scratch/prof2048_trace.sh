set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p2048t -o run -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 2 > $R/gpurun_out/p2048t.log 2>&1
cd $R
python3 tools/summarize_prof.py bygrid gpurun_out/p2048t/run_kernel_trace.csv gpurun_out/r04_p_bygrid_b2048.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/p2048t/run_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: print the sequence of the final 200 kernels with durations and gaps
last=rows[-215:]
prev=None
out=[]
for r in last:
    st,en=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(st-prev)/1e3 if prev else 0
    out.append("%-60s grid %6d dur %7.1f gap %6.1f" % (r["Kernel_Name"].split("(")[0][-60:], int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"])), (en-st)/1e3, gap))
    prev=en
open("gpurun_out/r04_p_seq_b2048.txt","w").write("\n".join(out))
PY
rm -f gpurun_out/p2048t/run_kernel_trace.csv
