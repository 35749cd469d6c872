#!/bin/bash
# Sanitizer runs of the HOST code (never on the GPU: GPU ASan is not available on this pool):   tools/sanitize.sh [log]
#   1. oracle (the checker, plain C + pthreads): ASan + UBSan under its golden tests; TSan on its renderer pool
#   2. host side of libntracer_hip.so (nt_api.cpp, nt_builder.cpp, nt_launch.cpp; kernels stubbed: tools/sanitize/kernel_stubs.cpp):
#      ASan + UBSan under the CPU test files that drive it (ABI, host API, builder, reference suite, pickling);
#      TSan on the threaded k-d builder
# Everything is built under build_ab/san/ (git-ignored).  Zero reports = every step prints OK.
log=${1:-profiles/r03_sanitizers.log}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build_ab/san
mkdir -p $out
cd $root
: > $log
say() { echo "$@" | tee -a $log; }
run() {  # name, command...: OK unless the command fails or a sanitizer report appears in its output
  name=$1; shift
  if "$@" > $out/$name.out 2>&1 && ! grep -q "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer\|ERROR: LeakSanitizer" $out/$name.out; then
    say "OK    $name: $(tail -1 $out/$name.out)"
  else
    say "FAIL  $name"; tail -30 $out/$name.out | tee -a $log
  fi
}
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
export UBSAN_OPTIONS=print_stacktrace=1
say "# $(date -u +%F) $(gcc --version | head -1); $(g++ --version | head -1)"
say "## oracle: -fsanitize=address,undefined"
gcc -O1 -g -std=c99 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $out/libntracer_oracle_asan.so oracle/ntracer_oracle.c -lm -lpthread || exit 1
run oracle_asan_golden env LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0 NTRACER_ORACLE_LIB=$out/libntracer_oracle_asan.so python3 -m pytest tests/test_oracle_golden.py -q -x -p no:cacheprovider
say "## oracle renderer pool: -fsanitize=thread"
gcc -O1 -g -std=c99 -ffp-contract=off -fsanitize=thread -o $out/oracle_pool_tsan tools/sanitize/oracle_pool_driver.c oracle/ntracer_oracle.c -lm -lpthread || exit 1
run oracle_pool_tsan $out/oracle_pool_tsan
say "## host side of libntracer_hip.so: -fsanitize=address,undefined (kernels stubbed)"
HOSTSRC="ntracer_amd/csrc/nt_api.cpp ntracer_amd/csrc/nt_builder.cpp ntracer_amd/csrc/nt_launch.cpp tools/sanitize/kernel_stubs.cpp"
g++ -O1 -g -std=c++17 -fPIC -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o $out/libntracer_host_asan.so $HOSTSRC -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib || exit 1
run host_asan_tests env LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0 NTRACER_HIP_LIB=$out/libntracer_host_asan.so NTRACER_HIP_SYSTEM_RUNTIME=1 \
    python3 -m pytest tests/test_abi.py tests/test_host_api.py tests/test_builder.py tests/test_reference_suite.py tests/test_pickle.py -q -x -m "not gpu" -p no:cacheprovider
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -o $out/builder_asan tools/sanitize/builder_driver.cpp ntracer_amd/csrc/nt_builder.cpp tools/sanitize/last_error_stub.cpp || exit 1
run builder_asan_driver env ASAN_OPTIONS=detect_leaks=1 $out/builder_asan 6000
say "## threaded k-d builder: -fsanitize=thread"
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -o $out/builder_tsan tools/sanitize/builder_driver.cpp ntracer_amd/csrc/nt_builder.cpp tools/sanitize/last_error_stub.cpp || exit 1
run builder_tsan_driver env NTRACER_BUILD_THREADS=6 $out/builder_tsan 6000
say "## done"
grep -c "^FAIL" $log | sed 's/^/failures: /' | tee -a $log
