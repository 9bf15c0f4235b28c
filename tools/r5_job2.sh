set -e
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_ipm.py -x -q -m gpu > gpurun_out/r5/job2_ipm_tests.log 2>&1 || { tail -60 gpurun_out/r5/job2_ipm_tests.log; exit 1; }
tail -3 gpurun_out/r5/job2_ipm_tests.log
python tools/ipm_probe.py > gpurun_out/r5/ipm_probe.jsonl 2> gpurun_out/r5/ipm_probe.err || { tail -20 gpurun_out/r5/ipm_probe.err; exit 1; }
cat gpurun_out/r5/ipm_probe.jsonl
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_ipm.py > gpurun_out/r5/job2_gpu_tests.log 2>&1 || { tail -60 gpurun_out/r5/job2_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r5/job2_gpu_tests.log
