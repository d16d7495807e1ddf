set -o pipefail
timeout -k 10 600 python -u -m pytest -x -q --timeout 60 --timeout-method thread tests/test_gpu_dp.py tests/test_gpu_models.py tests/test_gpu_sweep.py > gpurun_out/group_tests.log 2>&1 || { tail -40 gpurun_out/group_tests.log; exit 1; }
tail -2 gpurun_out/group_tests.log
for c in "cvt --graph" "cvt"; do CFG="$c" bash tools/diag/ab_cfg_env.sh VITK_DWCONV_SINK 2 || exit 1; done
