O=$(pwd)/gpurun_out
R=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py tests/test_gpu_distributed.py tests/test_gpu_api.py -q -k "wide or panel or Kg or gene_design or very" > $O/r5s_pytest_wide.log 2>&1
tail -4 $O/r5s_pytest_wide.log
timeout 900 python profiles/wide_ab.py --rounds 2 --steps 4 --cases 128:0,256:0,3:128,70:0 > $O/r5s_panels_at_c3.log 2>&1
tail -1 $O/r5s_panels_at_c3.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r5s -o t -- python3 $R/profiles/wide_ab.py --rounds 1 --steps 4 --cases 128:0,256:0,3:128 > $O/r5s_run.log 2>&1
f=$(find /tmp/r5s -name "*kernel_stats.csv" | head -1)
cp $f $O/r5s_panels_kernel_stats.csv
python3 - <<'P'
import csv
for r in csv.DictReader(open("/root/repo/gpurun_out/r5s_panels_kernel_stats.csv")):
    n=r["Name"]
    if "brie::" in n and float(r["AverageNs"])>1e5:
        print(n.split("(")[0][:72], r["Calls"], round(float(r["AverageNs"])/1e6,3), round(float(r["MinNs"])/1e6,3), round(float(r["MaxNs"])/1e6,3))
P
