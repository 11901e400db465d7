#!/bin/bash
# Sanitizer builds.  `bash tools/sanitize.sh cpu` runs what needs no GPU; `... gpu <outdir>` runs on the GPU box
# (gpurun -- 'bash tools/sanitize.sh gpu gpurun_out/<dir>').  Clean logs are kept under profiles/.
#   cpu:  oracle/stark_oracle.c (gcc) and tests/emu/ntt_emu.cpp (g++; the kernels' round bodies, the planner, field.cuh on the
#         host) under ASan + UBSan against their test suites; the HOST side of libstarkcore.so (hipcc -fsanitize=address,undefined:
#         device code is not instrumented) against the host-only suites: transcript, proof pickler, ABI
#   gpu:  the same ASan + UBSan library under the C-ABI parity tests and the Fri / FastStark host tests, tools/thread_stress.py;
#         a ThreadSanitizer build under tools/thread_stress.py (three prover threads in one process).
#         What cannot run under the preloaded ASan runtime, and why (profiles/r04/sanitize_gpu_notes.txt): anything that initialises
#         torch.cuda (torch's own dlopen of libcaffe2_nvrtc.so fails under ASan's dlopen interceptor: the sharded tests, the stream
#         join test), and tools/leak_check.py, which imports torch first (ROCm's ASan intercepts hsa_amd_memory_pool_allocate and reports
#         "out of memory" for a 4 MB allocation of the HIP RUNTIME itself, before the library is called) -- both run without ASan.
set -u
set -o pipefail
REPO=$(cd "$(dirname "$0")/.." && pwd); cd $REPO
MODE=${1:-cpu}; OUT=${2:-/tmp/sanitize}; mkdir -p $OUT
LIBS=${SAN_LIBS:-/tmp/sanitize_libs}; mkdir -p $LIBS   # the instrumented libraries (tens of MB) stay out of $OUT, which holds the logs
# `bash tools/sanitize.sh build` with SAN_LIBS=stark-anatomy_amd/.san builds the two instrumented libraries WITHOUT a GPU (hipcc cross-
# compiles); a later `gpu` run with the same SAN_LIBS finds them (newer than every source) and spends its GPU time on the tests only
CSRC=stark-anatomy_amd/csrc
CLANG_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
CLANG_TSAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so 2>/dev/null | head -1)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
build_lib() {   # $1 = sanitizer list, $2 = output
  if [ -f $2 ] && [ -z "$(find $CSRC include -newer $2 -type f | head -1)" ]; then echo "(using $2, built ahead of time)"; return 0; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=$1 -fno-omit-frame-pointer -shared-libsan \
      -Wno-unused-value -Wno-unused-result -Wno-option-ignored -shared -o $2 $CSRC/core.hip $CSRC/merkle_fri.hip $CSRC/polytree_geo.hip $CSRC/fourstep.hip 2>&1 | grep -E "error" ; test -f $2
}
status=0
if [ "$MODE" = build ]; then
  build_lib address,undefined $LIBS/libstarkcore_san.so || status=1
  if [ -n "$CLANG_TSAN" ]; then build_lib thread $LIBS/libstarkcore_tsan.so || status=1; fi
  ls -la $LIBS; echo "sanitize.sh build: status $status"; exit $status
fi
if [ "$MODE" = cpu ]; then
  gcc -O1 -g -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared -o $LIBS/libstark_oracle_san.so oracle/stark_oracle.c || status=1
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared -o $LIBS/libntt_emu_san.so tests/emu/ntt_emu.cpp || status=1
  GCC_RT="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
  echo "== oracle (gcc, ASan + UBSan): tests/test_oracle.py tests/test_polytree_model.py tests/test_geoseq_model.py"
  LD_PRELOAD=$GCC_RT STARK_ORACLE_LIB=$LIBS/libstark_oracle_san.so timeout 1500 python -m pytest tests/test_oracle.py tests/test_polytree_model.py tests/test_geoseq_model.py -x -q 2>&1 | tail -4 || status=1
  echo "== kernel emulation + planner (g++, ASan + UBSan): tests/test_emu.py"
  LD_PRELOAD=$GCC_RT STARK_ORACLE_LIB=$LIBS/libstark_oracle_san.so NTT_EMU_LIB=$LIBS/libntt_emu_san.so timeout 2400 python -m pytest tests/test_emu.py -x -q 2>&1 | tail -4 || status=1
  echo "== libstarkcore.so host side (hipcc, ASan + UBSan): tests/test_proof_pickle.py tests/test_host_cpu.py tests/test_abi.py"
  build_lib address,undefined $LIBS/libstarkcore_san.so || status=1
  LD_PRELOAD=$CLANG_RT STARKCORE_LIB=$LIBS/libstarkcore_san.so timeout 1500 python -m pytest tests/test_proof_pickle.py tests/test_host_cpu.py tests/test_abi.py -x -q 2>&1 | tail -4 || status=1
else
  echo "== libstarkcore.so host side under ASan + UBSan on the GPU"
  build_lib address,undefined $LIBS/libstarkcore_san.so || status=1
  for t in "tests/test_gpu_cabi.py -k 'not full_size and not big and not tunings and not stream_handle and not vec_wrap and not columns and not two_streams'" "tests/test_gpu_host.py" "tests/test_gpu_stark.py" "tests/test_gpu_geoseq.py -k 'not 1048'"; do
    echo "-- pytest $t"
    LD_PRELOAD=$CLANG_RT STARKCORE_LIB=$LIBS/libstarkcore_san.so timeout 1500 bash -c "set -o pipefail; python -m pytest $t -x -q -m gpu 2>&1 | tail -4" || status=1
  done
  echo "-- tools/leak_check.py (plain library: see the header)"; timeout 600 python tools/leak_check.py 2>&1 | grep -v "amdgpu.ids" | tail -3 || status=1
  echo "-- tools/thread_stress.py 10 3 (ASan)"; LD_PRELOAD=$CLANG_RT STARKCORE_LIB=$LIBS/libstarkcore_san.so timeout 600 python tools/thread_stress.py 10 3 2>&1 | tail -3 || status=1
  if [ -n "$CLANG_TSAN" ]; then
    echo "== ThreadSanitizer build under three prover threads"
    build_lib thread $LIBS/libstarkcore_tsan.so || status=1
    LD_PRELOAD=$CLANG_TSAN TSAN_OPTIONS="report_signal_unsafe=0:ignore_noninstrumented_modules=1:halt_on_error=0:exitcode=0" STARKCORE_LIB=$LIBS/libstarkcore_tsan.so \
        timeout 900 python tools/thread_stress.py 10 3 > $OUT/tsan_thread_stress.txt 2>&1; tail -3 $OUT/tsan_thread_stress.txt
    echo "ThreadSanitizer reports naming libstarkcore frames: $(grep -c 'libstarkcore_tsan' $OUT/tsan_thread_stress.txt)   (all reports: $(grep -c 'WARNING: ThreadSanitizer' $OUT/tsan_thread_stress.txt))"
  else
    echo "no ThreadSanitizer runtime in this toolchain"
  fi
fi
echo "sanitize.sh $MODE: status $status"
exit $status
