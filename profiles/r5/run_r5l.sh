O=$(pwd)/gpurun_out
R=$(pwd)
hipcc --offload-arch=gfx950 -O3 profiles/micro/mfma_wall.hip -o /tmp/mfma_wall && /tmp/mfma_wall > $O/r5l_mfma_wall.jsonl 2>&1
cat $O/r5l_mfma_wall.jsonl
timeout 900 python profiles/wide_ab.py --rounds 2 --steps 4 --cases 128:0,256:0,3:128,70:0 > $O/r5l_panels_at_c3.log 2>&1
tail -1 $O/r5l_panels_at_c3.log
