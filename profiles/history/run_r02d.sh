cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02e
python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -q -m gpu -k "async or streams or mixed or marginlik or loglik or lrt_reuses" > gpurun_out/r02e/pytest.log 2>&1
python profiles/e2e_fit_api.py > gpurun_out/r02e/e2e_prefetch.json 2> gpurun_out/r02e/err.log
python profiles/e2e_fit_api.py --no-prefetch > gpurun_out/r02e/e2e_noprefetch.json 2>> gpurun_out/r02e/err.log
python profiles/e2e_fit_api.py > gpurun_out/r02e/e2e_prefetch2.json 2>> gpurun_out/r02e/err.log
tail -3 gpurun_out/r02e/pytest.log; cat gpurun_out/r02e/*.json
