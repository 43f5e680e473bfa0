export HSA_ENABLE_IPC_MODE_LEGACY=0
for sw in "" "GLUE_KERNELS" "WGB_PREPACK"; do
python - "$sw" <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from grid_gcn_amd import train_ops
sw = sys.argv[1]
if sw: setattr(train_ops, sw, False)
import test_gpu_gridconv as t
try:
    t.test_graphed_train_step_equals_eager_step()
    print("off:", sw or "-", "PASS")
except AssertionError as e:
    print("off:", sw or "-", "FAIL", str(e)[:200])
PY
done
