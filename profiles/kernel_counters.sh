#!/bin/bash
# Counter evidence for ONE kernel: kernel trace + separate --pmc passes, summarised into one text file.
#   bash profiles/kernel_counters.sh <tag> <kernel-name-substring> <counter_workload.py args...>
# Passes (each its own run; --pmc never together with a trace domain):
#   1. --kernel-trace --stats                 -> average duration per launch
#   2. --pmc SQ set                           -> waves, wave cycles, VALU instructions / busy cycles, wait cycles
#   3. --pmc FETCH_SIZE                       -> HBM-side read bytes (x2 on gfx950 for wide coalesced reads)
#   4. --pmc WRITE_SIZE
#   5. (MFMA kernels) --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
set -u
TAG=$1; KERN=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/counters_$TAG
RAW=/tmp/brie_cnt_$TAG
rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
W="$R/profiles/counter_workload.py $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o t -- python3 $W > $OUT/workload_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d $RAW/sq -o sq -- python3 $W > $OUT/workload_sq.json 2> $OUT/sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -o f -- python3 $W > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/write -o w -- python3 $W > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM \
  --output-format csv -d $RAW/mix -o m -- python3 $W > /dev/null 2> $OUT/mix.err
python3 - "$KERN" "$RAW" "$OUT" "$TAG" <<'PY' > $OUT/summary.txt
import csv, glob, collections, json, sys
kern, raw, out, tag = sys.argv[1:5]
print("counter evidence %s -- kernel name contains %r" % (tag, kern))
try:
    print("workload:", open(out + "/workload_trace.json").read().strip().splitlines()[-1])
except Exception as e:
    print("workload line missing:", e)
dur = None
for f in glob.glob(raw + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Name"]:
            print("kernel-trace --stats: calls %s  avg %.3f ms  min %.3f  max %.3f   %s" % (
                r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Name"][:90]))
            dur = float(r["AverageNs"]) * 1e-9 if dur is None else dur
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(raw + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
d = {k: v / n for k, (n, v) in agg.items()}
print("per dispatch (rocprofv3 --pmc, separate passes):")
for k in sorted(d):
    print("  %-30s %16.6g   (dispatches %d)" % (k, d[k], agg[k][0]))
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    print("HBM traffic per launch: 2 x FETCH_SIZE + WRITE_SIZE = %.3f GB (FETCH x2: gfx950 tallies 128-B requests at 64 B)"
          % ((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / 1e9))
if "SQ_WAVE_CYCLES" in d:
    wc = d["SQ_WAVE_CYCLES"]
    for k, label in (("SQ_ACTIVE_INST_VALU", "VALU issuing"), ("SQ_ACTIVE_INST_ANY", "any instruction issuing"),
                     ("SQ_WAIT_ANY", "parked (s_waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "issue stall")):
        if k in d:
            print("share of wave cycles, %-28s %.3f" % (label + ":", d[k] / wc))
if dur and "SQ_ACTIVE_INST_VALU" in d:
    # SQ_ACTIVE_INST_VALU counts quad-cycles; 256 CUs x 4 SIMDs; nominal 2.4 GHz (profiled passes run lower: upper bound on time => lower bound)
    simd_cycles = dur * 2.4e9 * 1024
    print("VALU busy, share of all SIMD cycles (4 x SQ_ACTIVE_INST_VALU / (duration x 2.4 GHz x 1024 SIMDs)): %.3f"
          % (4 * d["SQ_ACTIVE_INST_VALU"] / simd_cycles))
    if "GRBM_GUI_ACTIVE" in d:
        print("effective clock under the profiler (GRBM_GUI_ACTIVE / duration): %.2f GHz  [if the counter is per device]"
              % (d["GRBM_GUI_ACTIVE"] / dur / 1e9))
if dur and "SQ_INSTS_VALU" in d:
    print("VALU wave-instructions per launch %.4g -> %.4g lane-instructions / s = %.3f of 3.93e13 (256 CU x 64 lanes x 2.4 GHz)"
          % (d["SQ_INSTS_VALU"], d["SQ_INSTS_VALU"] * 64 / dur, d["SQ_INSTS_VALU"] * 64 / dur / 3.93e13))
if dur and "SQ_INSTS_VALU_MFMA_MOPS_F32" in d:
    print("MFMA: %.4g FLOP per launch (MOPS x 512) = %.2f TFLOP/s; MFMA pipes busy %.3f of SIMD cycles"
          % (d["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512, d["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / dur / 1e12,
             d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (dur * 2.4e9 * 1024)))
PY
cat $OUT/summary.txt
