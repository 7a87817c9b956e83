# MFMA-kernel counters (one pass each; rocprofv3 --pmc with the program itself after --)
O=$(pwd)/gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/r5u -o c -- python3 $R/profiles/wide_ab.py --rounds 1 --steps 2 --cases 128:0 > $O/r5u_run.log 2>&1
f=$(find /tmp/r5u -name "*counter_collection.csv" | head -1)
python3 - "$f" > $O/r5u_mfma_counters.txt <<'P'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0][:60]
    if not any(x in k for x in ("fused_prior_mean","wide_design_grad","elbo_adam_step")): continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="GRBM_GUI_ACTIVE": n[k]+=1
for k,v in acc.items():
    d=max(n[k],1)
    print(k, "dispatches", n[k], {c: round(x/d) for c,x in v.items()})
    g=v.get("GRBM_GUI_ACTIVE",0)/d
    if g: print("   MFMA busy cycles per SIMD / kernel cycles:", round(v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/d/1024/g,3), " wait_any/wave_cycles", round(v.get("SQ_WAIT_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),3), " wait_inst/wave_cycles", round(v.get("SQ_WAIT_INST_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),3), " active/wave_cycles", round(v.get("SQ_ACTIVE_INST_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),3))
P
cat $O/r5u_mfma_counters.txt
