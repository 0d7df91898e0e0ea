#!/bin/bash
# Diagnostic build of one variant of the HIP library for same-box A/B timing (tools/ab_try.sh): only the 200-node instantiation
# (-DWRSN_ONLY_NPL=4), extra flags from the command line.   tools/ab_build.sh <name> [-D...]   ->  tools/lib_<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../multi_agent_rl_wrsn_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWRSN_ONLY_NPL=${WRSN_ONLY_NPL:-4} "$@" -o ../../tools/lib_$NAME.so wrsn_api.hip
