#!/bin/bash
# Diagnostic build of the HIP library (-DDG_PROF: per-phase and per-scenario cycle counters), always fresh, next to the product library.
set -e
cd "$(dirname "$0")/../dgsqp_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDG_PROF -o libdgsqp_hip_prof.so dgsqp_api.hip
ls -la libdgsqp_hip_prof.so
