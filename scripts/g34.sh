mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout=600 > gpurun_out/g34_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g34_pytest.log)
tail -3 gpurun_out/g34_pytest.log
SLUAMD_LIB=$GRAFT_REPO_ROOT/superlu_dist_amd/libsluamd_stamps.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 2>&1 >/dev/null | grep "k_diag_lu2" 
bash scripts/ab.sh g34 $GRAFT_REPO_ROOT/superlu_dist_amd/libsluamd_old.so $GRAFT_REPO_ROOT/superlu_dist_amd/libsluamd_old.so
