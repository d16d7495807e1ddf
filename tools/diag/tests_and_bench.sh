set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest -x -q --timeout 180 --timeout-method thread tests -m gpu > gpurun_out/full_gpu_tests.log 2>&1 || { tail -30 gpurun_out/full_gpu_tests.log; exit 1; }
tail -2 gpurun_out/full_gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
tail -c 3000 gpurun_out/bench_default.log
