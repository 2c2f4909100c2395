#!/bin/bash
# Register / scratch / occupancy table of every kernel in a .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
#   tools/kernel_resources.sh collaborative-distillation_amd/csrc/conv3x3_f16.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/remark: Function Name:/{n=$5} /remark: +VGPRs:/{v=$(NF-1)} /remark: +AGPRs:/{a=$(NF-1)} /ScratchSize/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} /VGPRs Spill/{sp=$(NF-1)} /LDS Size/{print n, "vgpr", v, "agpr", a, "scratch", s, "spill", sp, "occ", o}' | c++filt | sed 's/(anonymous namespace):://'
