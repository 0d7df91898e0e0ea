#!/bin/bash
# Diagnostic build of the HIP library with in-kernel phase timers (never the product build).
set -e
cd "$(dirname "$0")/../multi_agent_rl_wrsn_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWRSN_PROFILE=${WRSN_PROFILE_LEVEL:-1} -o ../../tools/libwrsn_hip_profile${WRSN_PROFILE_SUFFIX:-}.so wrsn_api.hip
