#!/bin/bash
# tools/sanitize_cpu.sh -- AddressSanitizer + UBSan over everything that runs on the CPU: the oracle, the input generators and
# the host code of the product library (resolver, Reed-Solomon host build, C ABI argument paths).  GPU ASAN is not available
# on this pool; the device code is covered by the parity and fuzz suites instead.
set -e
root=$(cd "$(dirname "$0")/.." && pwd); tmp=$(mktemp -d)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared -pthread -o $tmp/liboracle1090.so $root/oracle/oracle1090.c $root/oracle/oracle978.c $root/oracle/expected1090.c $root/oracle/oracle2400.c -lm
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared -pthread -o $tmp/libadsb_synth.so $root/libadsb_amd/csrc/synth1090.c $root/libadsb_amd/csrc/synth978.c -lm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++20 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-option-ignored \
    -I$root/include -I$root/libadsb_amd/csrc -o $tmp/libadsb_amd.so $root/libadsb_amd/csrc/*.hip $root/libadsb_amd/csrc/*.cpp
cp $root/oracle/liboracle1090.so $tmp/oracle.bak; cp $root/libadsb_amd/libadsb_synth.so $tmp/synth.bak
restore() { cp $tmp/oracle.bak $root/oracle/liboracle1090.so; cp $tmp/synth.bak $root/libadsb_amd/libadsb_synth.so; touch $root/oracle/liboracle1090.so $root/libadsb_amd/libadsb_synth.so; rm -rf $tmp; }
trap restore EXIT
cp $tmp/liboracle1090.so $root/oracle/liboracle1090.so; cp $tmp/libadsb_synth.so $root/libadsb_amd/libadsb_synth.so; touch $root/oracle/liboracle1090.so $root/libadsb_amd/libadsb_synth.so
cd $root
echo "== oracle + generators under gcc ASAN/UBSan"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python -m pytest tests/test_oracle_golden.py tests/test_uat978_cpu.py tests/test_mode2400_cpu.py -x -q
cp $tmp/oracle.bak $root/oracle/liboracle1090.so; cp $tmp/synth.bak $root/libadsb_amd/libadsb_synth.so; touch $root/oracle/liboracle1090.so $root/libadsb_amd/libadsb_synth.so
echo "== product host code under clang ASAN/UBSan"
ASAN_OPTIONS=detect_leaks=0 ADSB_AMD_LIB=$tmp/libadsb_amd.so LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) python -m pytest tests/test_capi_cpu.py tests/test_uat978_cpu.py -x -q
