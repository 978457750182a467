#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over everything of this repository that runs on a CPU (GPU ASan is not available on this pool):
#   1. the asset layer (assets.cpp + godot_import.cpp, g++ host build) under the fuzz of tests/test_asset_fuzz.py (truncated / bit-flipped / header-mangled
#      BMP, TGA, .ctex, .ctex3d; hostile sizes and levels), 3 seeds x 400 mutations x 6 files
#   2. the C oracle under tests/test_oracle_structure.py, test_oracle_golden.py, test_compositor.py
#   3. the kernel cores compiled for the host (tests/hostsim) under tests/test_hostsim_core.py
# Sanitized builds replace the normal test libraries for the run and are restored afterwards.  usage: bash tools/sanitize_cpu.sh
set -u
R=$(cd "$(dirname "$0")/.." && pwd); W=$(mktemp -d); cd $R
PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
SAN="-g -O1 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer"
echo "== 1. asset layer"
g++ -std=c++17 $SAN -I$R/include -o $W/libassets_asan.so godot-volumetric-cloud-demo-v2_amd/csrc/assets.cpp godot-volumetric-cloud-demo-v2_amd/csrc/godot_import.cpp -lpthread || exit 1
cat > $W/fuzz_asan.py <<PY
import ctypes as C, re, sys, tempfile, textwrap
src = open("$R/tests/test_asset_fuzz.py").read()
child = textwrap.dedent(re.search(r"CHILD = textwrap.dedent\('''(.*?)'''\) % \(ROOT,\)", src, re.S).group(1)) % ("$R",)
child = child.replace("import gvcd_amd\nL = gvcd_amd.lib()", "L = C.CDLL('$W/libassets_asan.so')\nL.csky_assets_last_error.restype = C.c_char_p\nL.csky_mip_offset.restype = C.c_size_t")
sys.argv = ["x", tempfile.mkdtemp(), sys.argv[1], sys.argv[2]]
exec(child)
PY
for seed in 1 2 3; do LD_PRELOAD=$PRE python $W/fuzz_asan.py $seed 400 2>&1 | tail -3; done
echo "== 2. oracle"
cp oracle/libcskoracle.so $W/o.so
gcc -std=c11 -ffp-contract=off -fno-fast-math -fopenmp $SAN -o oracle/libcskoracle.so oracle/cloudsky_oracle.c -lm
LD_PRELOAD=$PRE python -m pytest tests/test_oracle_structure.py tests/test_oracle_golden.py tests/test_compositor.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
cp $W/o.so oracle/libcskoracle.so; touch oracle/libcskoracle.so
echo "== 3. kernel cores on the host"
make -C tests/hostsim -s; cp tests/hostsim/libhostsim.so $W/h.so
(cd tests/hostsim && g++ -std=c++17 -ffp-contract=off -fno-fast-math $SAN -o libhostsim.so hostsim.cpp)
LD_PRELOAD=$PRE python -m pytest tests/test_hostsim_core.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
cp $W/h.so tests/hostsim/libhostsim.so; touch tests/hostsim/libhostsim.so
rm -rf $W
