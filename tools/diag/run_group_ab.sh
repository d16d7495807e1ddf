set -o pipefail
timeout -k 10 600 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_kernels.py -k "attention" > gpurun_out/group_tests.log 2>&1 || { tail -40 gpurun_out/group_tests.log; exit 1; }
tail -2 gpurun_out/group_tests.log
timeout -k 10 600 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_sweep.py tests/test_gpu_models.py -k "attention or coatnet or attn" > gpurun_out/group_tests2.log 2>&1 || { tail -40 gpurun_out/group_tests2.log; exit 1; }
tail -2 gpurun_out/group_tests2.log
CFG="coatnet" bash tools/diag/ab_cfg_env.sh VITK_ATTN_FLASH_ONE_PASS 2
