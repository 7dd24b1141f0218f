mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout=600 > gpurun_out/g32_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g32_pytest.log)
tail -3 gpurun_out/g32_pytest.log
bash scripts/ab.sh g32 $GRAFT_REPO_ROOT/superlu_dist_amd/libsluamd_old.so $GRAFT_REPO_ROOT/superlu_dist_amd/libsluamd_old.so
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks.json 2> /tmp/ks.err
cd $GRAFT_REPO_ROOT
db=$(find /tmp/ks -name "*.db" | head -1)
python scripts/rocpd_stats.py $db 2>&1 | grep -E "k_diag_lu2|k_panel_trsm"
python scripts/timeline.py $db 2 2>&1 | tail -8
