# A debug build of the engine + driver (symbols, -rdynamic, no sanitizer) under glibc's allocator checks, with a call stack on abort.
O=$1; shift; mkdir -p $O
S=finmath-lib-cuda-extensions_amd/csrc; H=finmath-lib-cuda-extensions_amd/host; B=/tmp/fmdebug; mkdir -p $B
CXX="/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-omit-frame-pointer -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Ifinmath-lib-cuda-extensions_amd/build"
for f in runtime abi mersenne jit sharded; do $CXX -c $S/$f.cpp -o $B/$f.o || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -rdynamic -o $B/lmm_debug $B/runtime.o $B/abi.o $B/mersenne.o $B/jit.o $B/sharded.o finmath-lib-cuda-extensions_amd/build/kernels.o \
   -x c++ -O1 -g -std=c++17 -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $H/lmm_hip_main.cpp -lhiprtc -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,/opt/rocm/lib || exit 1
FMHIP_BACKTRACE=1 LD_PRELOAD=libc_malloc_debug.so.0 MALLOC_CHECK_=3 timeout -k 10 900 $B/lmm_debug "$@" > $O/debug.json 2> $O/debug.err
echo "rc $?"; tail -c 5000 $O/debug.err
