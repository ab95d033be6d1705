#!/usr/bin/env python3
"""Lists the reference's libhighs translation units (paths relative to <reference>/highs) by reading the source lists of
its own cmake/sources.cmake — the CMake-free recipe of integration/Makefile compiles exactly what the reference's build
compiles (default configuration: CPU cuPDLP-C, no HiPO, no CUDA), minus the TUs the drop-in replaces."""
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
txt = open(os.path.join(REF, "cmake", "sources.cmake")).read()
# the lists the default (FAST_BUILD, HIPO off, CUPDLP_CPU) libhighs is made of: highs/CMakeLists.txt adds basiclu, ipx,
# cupdlp, the hipo interface layers and highs sources; amd/metis/rcm/blas only with -DHIPO=ON, cuda_sources only with
# CUPDLP_GPU (checked against the object list of a CMake build of the reference: 200 TUs besides the replaced ones)
WANT = ["basiclu_sources", "ipx_sources", "cupdlp_sources", "hipo_sources", "factor_highs_sources", "hipo_util_sources",
        "highs_sources"]
# what the drop-in replaces: the cuPDLP-C wrapper + vendored C, the HiPDLP wrapper + hipdlp/*, the MPS reader TU
DROP = re.compile(r"^(pdlp/CupdlpWrapper\.cpp|pdlp/cupdlp/|pdlp/HiPdlpWrapper\.cpp|pdlp/hipdlp/|io/FilereaderMps\.cpp)")
# --dropped: exactly those TUs instead — what the UNMODIFIED reference library (integration/Makefile libhighs_reference.so.1:
# the goldens' generator and the CPU baseline of Highs::run()) adds to the common object list
only_dropped = len(sys.argv) > 2 and sys.argv[2] == "--dropped"
out = []
for name in WANT:
    m = re.search(r"set\(%s\s+(.*?)\)" % name, txt, re.S)
    for tok in m.group(1).split():
        if re.search(r"\.(c|cc|cpp)$", tok) and bool(DROP.search(tok)) == only_dropped and tok not in out:
            out.append(tok)
keep_reader = len(sys.argv) > 2 and sys.argv[2] == "--with-reader"
if keep_reader:
    out.append("io/FilereaderMps.cpp")
print("\n".join(out))
