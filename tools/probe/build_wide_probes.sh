#!/bin/bash
# builds tools/probe/wide_probe_<name> for each "name:flags" argument (run here or on the GPU box)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -I include -I panoptikon_amd/csrc $flags -o tools/probe/wide_probe_$name tools/probe/wide_probe.hip 2>&1 | grep -E "error" -A3
done
