#!/bin/bash
# Drop-in proof (build container only): link a libhighs.so from the REFERENCE's own object
# files with the PDLP wrapper TU + vendored cuPDLP-C objects replaced by
# integration/CupdlpWrapperMi355x.cpp / HiPdlpWrapperMi355x.cpp -> libpdlp_mi355x.so.  The reference's unmodified
# bin/highs and bin/unit_tests then run the MI355X path (LD_LIBRARY_PATH picks this libhighs).
#   $REF_BUILD : an existing CPU build of the reference (object files are reused; its build
#                system is NOT run here).  Outputs go to integration/_build/ (git-ignored; the
#                directory travels to the GPU box with gpurun).
set -euo pipefail
REF=${REF:-/root/reference}
REF_BUILD=${REF_BUILD:-/tmp/ref_build}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
OUT=$HERE/_build
mkdir -p "$OUT"
for w in CupdlpWrapperMi355x HiPdlpWrapperMi355x FilereaderMpsMi355x; do
  g++ -std=c++17 -O2 -fPIC -I"$REF/highs" -I"$REF_BUILD" -I"$ROOT/include" -c "$HERE/$w.cpp" -o "$OUT/$w.o"
done
# the reference's MPS reader TU stays in the build under another class name: FilereaderMpsMi355x.cpp calls it for what the
# library's reader does not take on (fixed-column files, writing); compiled from the source where it lies
g++ -std=c++17 -O2 -fPIC -DFilereaderMps=FilereaderMpsReference -I"$REF/highs" -I"$REF_BUILD" -I"$REF/extern" -I"$REF/extern/zstr" \
    -c "$REF/highs/io/FilereaderMps.cpp" -o "$OUT/FilereaderMpsReference.o"
# both reference PDLP paths are left out: cuPDLP-C (wrapper + vendored C) and HiPDLP (wrapper + hipdlp/*.cc),
# and the MPS file reader TU (replaced by FilereaderMpsMi355x.cpp -> pdlp_mi355x_read_mps)
OBJS=$(find "$REF_BUILD/highs/CMakeFiles/highs.dir" -name '*.o' | grep -v -E 'pdlp/CupdlpWrapper\.cpp\.o|pdlp/cupdlp/|pdlp/HiPdlpWrapper\.cpp\.o|pdlp/hipdlp/|io/FilereaderMps\.cpp\.o')
/opt/rocm/lib/llvm/bin/clang++ -flto=thin -fuse-ld=lld -O3 -shared -o "$OUT/libhighs.so.1" -Wl,-soname,libhighs.so.1 $OBJS "$OUT/CupdlpWrapperMi355x.o" "$OUT/HiPdlpWrapperMi355x.o" "$OUT/FilereaderMpsMi355x.o" "$OUT/FilereaderMpsReference.o" \
    -L"$ROOT/highs_amd/lib" -lpdlp_mi355x -Wl,-rpath,'$ORIGIN/../../highs_amd/lib' -lz -lpthread -ldl
# the same library with the REFERENCE's MPS reader TU left in: what tests/golden/make_golden_mps.py and tools/mps_bench.py
# load to see / time the reference's own reader (io/FilereaderMps.cpp -> HMpsFF.cpp), never the product
OBJS_REFIO=$(find "$REF_BUILD/highs/CMakeFiles/highs.dir" -name '*.o' | grep -v -E 'pdlp/CupdlpWrapper\.cpp\.o|pdlp/cupdlp/|pdlp/HiPdlpWrapper\.cpp\.o|pdlp/hipdlp/')
/opt/rocm/lib/llvm/bin/clang++ -flto=thin -fuse-ld=lld -O3 -shared -o "$OUT/libhighs_ref_reader.so" -Wl,-soname,libhighs_ref_reader.so $OBJS_REFIO "$OUT/CupdlpWrapperMi355x.o" "$OUT/HiPdlpWrapperMi355x.o" \
    -L"$ROOT/highs_amd/lib" -lpdlp_mi355x -Wl,-rpath,'$ORIGIN/../../highs_amd/lib' -lz -lpthread -ldl
# an unmodified C client of the reference's C API (Highs_create / Highs_passLp / Highs_run / ...)
gcc -O2 -I"$REF/highs" -I"$REF_BUILD" "$HERE/capi_check.c" -o "$OUT/capi_check" -L"$OUT" -l:libhighs.so.1 -lm \
    -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,"$ROOT/highs_amd/lib"
cp "$REF_BUILD/bin/highs" "$OUT/highs_ref_cli"
cp "$REF_BUILD/bin/unit_tests" "$OUT/unit_tests_ref"
echo "built $OUT/libhighs.so.1 ; run with LD_LIBRARY_PATH=$OUT"
nm -D "$OUT/libhighs.so.1" | grep -c pdlp_mi355x_solve
