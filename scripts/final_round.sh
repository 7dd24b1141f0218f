#!/bin/bash
# End-of-round GPU run (from the repo root on the GPU box): the whole -m gpu suite, then scripts/final_measure.sh (PMC traffic, kernel statistics, timelines, default bench line, complex16).
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/r06_gputest_final.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_final.log)
tail -4 gpurun_out/r06_gputest_final.log
bash scripts/final_measure.sh r06 > gpurun_out/r06_final_measure.log 2>&1
tail -5 gpurun_out/r06_final_measure.log
cp profiles/r06_pmc_schur.json gpurun_out/r06_pmc_schur_copy.json
