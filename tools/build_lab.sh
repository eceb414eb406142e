#!/bin/bash
# build the GEMM lab (diagnostic binary; not part of the product library) with its assembly kept next to it
set -e
cd "$(dirname "$0")/_build" 2>/dev/null || { mkdir -p "$(dirname "$0")/_build"; cd "$(dirname "$0")/_build"; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps=obj ../gemm_lab.hip -o gemm_lab -ldl
