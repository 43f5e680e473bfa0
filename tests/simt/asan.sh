#!/bin/bash
# AddressSanitizer over the product's kernels on the CPU build:  tests/simt/asan.sh [pytest arguments]
#   default: the index operators + the attention kernels (tests/test_simt_index.py, tests/test_simt_train.py)
# The python process itself is not instrumented: the ASan runtime is preloaded, leak checking off.
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$(dirname "$0")/../.."
python tests/simt/build.py --asan > /dev/null || exit 1
export GG_SIMT_ASAN=1
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:halt_on_error=1:allocator_may_return_null=1
LD_PRELOAD=$RT python -m pytest -x -q -p no:cacheprovider ${@:-tests/test_simt_index.py tests/test_simt_train.py}
