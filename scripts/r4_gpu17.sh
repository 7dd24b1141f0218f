#!/bin/bash
mkdir -p gpurun_out
bash scripts/env_ab.sh g17 "SLUAMD_NO_TILE_MAPS=1" "SLUAMD_NO_TILE_MAPS=1 SLUAMD_NO_MERGE_TILES=1"
