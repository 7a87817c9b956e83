cd $GRAFT_REPO_ROOT; O=gpurun_out/r03n; mkdir -p $O
X="--no-pmc --no-f32-leg --no-e2e"
for cfg in "--config c3" "--config c3 --mc 3" "--config c2" "--config c2 --mc 3"; do
  echo "== $cfg" | tee -a $O/ab_mw.log
  bash profiles/ab.sh 3 "$X $cfg" base mw8 2>&1 | tee -a $O/ab_mw.log
done
python profiles/occ_ab.py 2>/dev/null | tee $O/occ_ab_after.log
