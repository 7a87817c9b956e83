python bench.py --steps 6000 --warmup 3 --no-pmc --no-e2e --no-cpu-baseline --no-psi-check --no-f32-leg > gpurun_out/r8v_neighbour.json 2> gpurun_out/r8v_neighbour.err &
NB=$!
sleep 45
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_parity.py -k "many_steps or fused_launch" -x -q 2>&1 | tail -1; done
SHAPES=200x500x0x2,200x1500x0x2 STEPS=300 timeout 300 python profiles/fused_steps.py 2>&1 >/dev/null | grep shape
kill -0 $NB 2>/dev/null && echo "neighbour still running" || echo "neighbour had ended"
kill $NB 2>/dev/null; wait $NB 2>/dev/null
echo done
