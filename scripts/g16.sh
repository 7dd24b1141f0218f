mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/g16_gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g16_gputest.log)
tail -5 gpurun_out/g16_gputest.log
