# Host-side AddressSanitizer build of the engine (device code unchanged) + the native driver, run on a real GPU: for faults that only
# show under real memory pressure.  $1 = output dir, rest = lmm_hip arguments
O=$1; shift; mkdir -p $O
S=finmath-lib-cuda-extensions_amd/csrc; H=finmath-lib-cuda-extensions_amd/host; B=/tmp/fmasan; mkdir -p $B
CXX="/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-omit-frame-pointer -fsanitize=address -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Ifinmath-lib-cuda-extensions_amd/build"
for f in runtime abi mersenne jit sharded; do $CXX -c $S/$f.cpp -o $B/$f.o || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -shared-libasan -o $B/lmm_asan_gpu $B/runtime.o $B/abi.o $B/mersenne.o $B/jit.o $B/sharded.o finmath-lib-cuda-extensions_amd/build/kernels.o \
   -x c++ -O1 -g -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $H/lmm_hip_main.cpp -x none -lhiprtc -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,/opt/rocm/lib || exit 1
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
LD_LIBRARY_PATH=$(dirname $(/opt/rocm/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)):$LD_LIBRARY_PATH timeout -k 10 900 $B/lmm_asan_gpu "$@" > $O/asan.json 2> $O/asan.err
echo "rc $?"; tail -c 6000 $O/asan.err
