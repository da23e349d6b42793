#!/bin/bash
# Host-side sanitizer run, CPU container only: builds the HOST half of csrc/ with AddressSanitizer + UBSan
# (make asan -> build/asan/libimt_hip_asan.so) and runs the C-ABI tests and the argument-validation tests against it.
# The ASan runtime must be the first library of the (uninstrumented) python process, hence LD_PRELOAD; leak checking is off
# (CPython and torch leak by design at exit).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C $root/imagetranslate_amd/csrc asan -j8 > /dev/null
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $root
IMT_LIB=$root/build/asan/libimt_hip_asan.so LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python3 -m pytest tests/test_cabi.py tests/test_host_validation.py -x -q -p no:cacheprovider "$@"
