export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/s10
rm -f gpurun_out/s10/par.txt
for i in 1 2 3 4 5 6; do
  GG_PARITY_REPORT=$PWD/gpurun_out/s10/par.txt timeout 600 python -m pytest tests/test_gpu_gridconv.py tests/test_model_cls.py -q -m gpu -k "ragged or full_size or gridify_up_variant or cls_model or stock_modules or fwd_bwd" > gpurun_out/s10/t$i.log 2>&1; tail -1 gpurun_out/s10/t$i.log
done
grep model gpurun_out/s10/par.txt | sort | uniq -c | sort -rn | head -40
