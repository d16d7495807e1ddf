set -o pipefail
timeout -k 10 600 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_models.py tests/test_gpu_sweep.py -k "mbconv or coatnet or batchnorm" > gpurun_out/group_tests.log 2>&1 || { tail -40 gpurun_out/group_tests.log; exit 1; }
tail -2 gpurun_out/group_tests.log
timeout -k 10 600 env VITK_BN_ACT_FUSED=0 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_models.py -k "mbconv" > gpurun_out/group_tests0.log 2>&1 || { tail -40 gpurun_out/group_tests0.log; exit 1; }
tail -1 gpurun_out/group_tests0.log
CFG="coatnet" bash tools/diag/ab_cfg_env.sh VITK_BN_ACT_FUSED 2
