mkdir -p gpurun_out
free -g | head -2; nproc
mem=$(free -g | awk '/^Mem:/{print $7}')
if [ "$mem" -lt 400 ]; then echo "host memory $mem GB: too small for the 300^3 planner run"; exit 0; fi
export SLUAMD_LIB=$GRAFT_REPO_ROOT/oracle/libsluamd_emul.so SLUAMD_EMUL_LAZY_ZERO=1 OMP_NUM_THREADS=16
( time timeout 1000 python scripts/grid_footprint_check.py 300 2 2 2 --create-only ) > gpurun_out/r05_capacity_300_planner.txt 2>&1
tail -14 gpurun_out/r05_capacity_300_planner.txt
