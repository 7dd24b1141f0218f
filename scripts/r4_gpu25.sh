#!/bin/bash
mkdir -p gpurun_out
bash scripts/env_ab.sh g25 SLUAMD_FUSE_TAIL_OFF=10 SLUAMD_FUSE_TAIL_OFF=20 SLUAMD_FUSE_TAIL_OFF=40
