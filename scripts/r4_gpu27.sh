#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04_bench100_final.json 2> gpurun_out/r04_bench100_final.err
tail -c 400 gpurun_out/r04_bench100_final.json
