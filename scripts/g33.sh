mkdir -p gpurun_out
SLUAMD_LIB=$GRAFT_REPO_ROOT/superlu_dist_amd/libsluamd_stamps.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 2>&1 >/dev/null | grep "k_diag_lu2" 
SLUAMD_NO_LOOKAHEAD=1 SLUAMD_LIB=$GRAFT_REPO_ROOT/superlu_dist_amd/libsluamd_stamps.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 2>&1 >/dev/null | grep "k_diag_lu2" 
