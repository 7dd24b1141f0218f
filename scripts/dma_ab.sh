#!/bin/bash
# round 6: LDS-DMA loader of the clean sources -- parity subset, then same-box A/B against the register-staged clean loader and the build without the destination touch
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_grid.py -q --timeout=300 -x > gpurun_out/dma_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/dma_pytest.log)
grep -E "passed|failed|FAILED|rc " gpurun_out/dma_pytest.log | tail -8
bash scripts/ab.sh dma ab/libsluamd_nodma.so ab/libsluamd_notouch.so ab/libsluamd_nodma_notouch.so 2>&1 | tee gpurun_out/dma_ab.txt
