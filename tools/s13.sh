export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/s13
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_gridconv.py -x -q -m gpu -k "att_bn2 or att_pairmax or att_fwd_noz" > gpurun_out/s13/tests.log 2>&1; tail -5 gpurun_out/s13/tests.log
bash tools/s12.sh
