#!/bin/bash
mkdir -p gpurun_out
bash scripts/solve_profile.sh g34 SLUAMD_SWEEP_WIDE_V=5
