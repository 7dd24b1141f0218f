#!/bin/bash
mkdir -p gpurun_out
bash scripts/ab.sh g7 $PWD/ab/libsluamd_unpacked.so
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks.json 2> /tmp/ks.err
cd $R
python scripts/rocpd_stats.py $(find /tmp/ks -name "*.db" | head -1) > gpurun_out/g7_kernel_stats_packed.txt 2>&1
head -12 gpurun_out/g7_kernel_stats_packed.txt
cd /tmp && rm -rf /tmp/ks2
SLUAMD_LIB=$R/ab/libsluamd_unpacked.so rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks2.json 2> /tmp/ks2.err
cd $R
python scripts/rocpd_stats.py $(find /tmp/ks2 -name "*.db" | head -1) > gpurun_out/g7_kernel_stats_unpacked.txt 2>&1
head -12 gpurun_out/g7_kernel_stats_unpacked.txt
