SLUAMD_SOLVE_DEBUG=1 SLUAMD_ZFUSE_MAX_NODES=16 python bench.py --workload zgrid2d --n 1000 --steps 3 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "host enqueue" | tail -3
SLUAMD_SOLVE_DEBUG=1 SLUAMD_ZFUSE_MAX_NODES=0 python bench.py --workload zgrid2d --n 1000 --steps 3 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "host enqueue" | tail -3
SLUAMD_SOLVE_DEBUG=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 2>&1 >/dev/null | grep "host enqueue" | tail -3
