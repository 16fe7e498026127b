#!/bin/bash
# Host-side sanitizer pass (no GPU needed; GPU ASan is not available on this pool): the pure-host translation units are
# rebuilt with -fsanitize=address,undefined, linked with the regular device objects into /tmp/asan_lib, and the CPU tests of
# the container / JSON / pairing code plus the mutation fuzzer run against that library.  Run from the repository root
# after `make`.
set -e
ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p build/obj_asan/prover /tmp/asan_lib
for f in runtime.cpp host_ffi.cpp prover/prover.cpp prover/pairing.cpp prover/containers.cpp prover/cache.cpp prover/assemble.cpp prover/multi.cpp; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Iinclude -Iicicle-snark_amd/csrc -Wno-unused-result -Wno-option-ignored \
    -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -x hip -c icicle-snark_amd/csrc/$f -o build/obj_asan/$f.o
done
OBJS="$(find build/obj -name '*.o' | grep -v -E 'runtime.cpp.o|host_ffi.cpp.o|prover/prover.cpp.o|prover/pairing.cpp.o|prover/containers.cpp.o|prover/cache.cpp.o|prover/assemble.cpp.o|prover/multi.cpp.o') $(find build/obj_asan -name '*.o')"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o /tmp/asan_lib/libicicle_snark_hip.so $OBJS -lpthread
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/real_lib.so
trap 'cp /tmp/real_lib.so $L' EXIT
cp /tmp/asan_lib/libicicle_snark_hip.so $L
export LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_abi.py tests/test_verify.py -x -q -m "not gpu"
python scratch/fuzz_containers.py 1 3000
python scratch/fuzz_containers.py 2 3000
