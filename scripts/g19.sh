mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -q -x --timeout=600 > gpurun_out/g19_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g19_pytest.log)
tail -3 gpurun_out/g19_pytest.log
