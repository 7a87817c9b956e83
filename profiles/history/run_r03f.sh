cd $GRAFT_REPO_ROOT; O=gpurun_out/r03f; mkdir -p $O
X="--no-pmc --no-f32-leg --no-e2e"
for cfg in "--config c2 --mc 3" "--config c2" "--config c3 --mc 3" "--config c3" "--config c2 --mc 3 --kc 5" "--config c3 --kc 12"; do
  echo "== $cfg" | tee -a $O/ab_w2.log
  bash profiles/ab.sh 2 "$X $cfg" base w2 2>&1 | tee -a $O/ab_w2.log
done
