#!/bin/bash
# Variant library with ONE csrc file recompiled under extra defines (ablations, tuning constants):
#   tools/micro/mkvar.sh <name> <file.hip> <defines...>  ->  grid_gcn_amd/lib/libgridgcn_hip_<name>.so
# (run where the regular library has been built: the other objects come from lib/obj; select the variant at
#  run time with GG_HIP_LIB=<path>).  Examples: the dW ablations of DESIGN 3.5 (f)
#   tools/micro/mkvar.sh a1 gridgcn_direct.hip -DGG_DW_ABLATE=1     # no MFMAs
#   tools/micro/mkvar.sh a2 gridgcn_direct.hip -DGG_DW_ABLATE=2     # no loads in the main rounds
#   tools/micro/mkvar.sh d8 gridgcn_direct.hip -DGG_DW_D16=8        # register sets of the 16-tile dW form
cd "$(dirname "$0")/../../grid_gcn_amd" || exit 1
n=$1; f=$2; shift 2
o=/tmp/ggvar_${n}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c csrc/$f -o $o || exit 1
objs=$(ls lib/obj/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $o -o lib/libgridgcn_hip_$n.so && echo "built lib/libgridgcn_hip_$n.so"
