set -e
mkdir -p gpurun_out/r5
python tools/ubench/issue.py > gpurun_out/r5/issue_ubench.txt 2>&1
tail -20 gpurun_out/r5/issue_ubench.txt
python -m pytest tests/test_gpu_n7_probe.py tests/test_gpu_actor.py tests/test_gpu_dist.py "tests/test_gpu_config_fuzz.py::test_random_barrier_family_is_bit_exact" -x -q -m gpu -s > gpurun_out/r5/job1_tests.log 2>&1 || { tail -50 gpurun_out/r5/job1_tests.log; exit 1; }
tail -5 gpurun_out/r5/job1_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_driverlike_0.json 2> gpurun_out/r5/bench_driverlike_0.err
cut -c1-600 gpurun_out/r5/bench_driverlike_0.json
