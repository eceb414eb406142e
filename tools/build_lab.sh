#!/bin/bash
# build the GEMM labs (diagnostic binaries; not part of the product library) with their assembly kept next to them
#   gemm_lab: native fp32 MFMA tile candidates with cycle stamps      x3_lab: split-operand (bf16x3) accuracy + tile lab
set -e
cd "$(dirname "$0")/_build" 2>/dev/null || { mkdir -p "$(dirname "$0")/_build"; cd "$(dirname "$0")/_build"; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps=obj ../gemm_lab.hip -o gemm_lab -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps=obj ../x3_lab.hip -o x3_lab -ldl
