mkdir -p gpurun_out
(SLUAMD_SOLVE_GROUPS=1 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/g41_gputest_groups.log 2>&1; echo "pytest rc $?" >> gpurun_out/g41_gputest_groups.log)
tail -4 gpurun_out/g41_gputest_groups.log
