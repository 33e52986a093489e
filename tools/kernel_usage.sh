#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_usage.sh csvplus_amd/csrc/chain.hip [name filter]
f="$1"; pat="${2:-.}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/remark: Function Name:/ {name=$(NF-1)} / VGPRs: / {v=$(NF-1)} / AGPRs: / {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)} /LDS Size/ {l=$(NF-1); print name, "vgpr="v, "agpr="a, "scratch="s, "vspill="sp, "occ="o, "lds="l}' |
  c++filt | sed 's/(cph::[^)]*)//; s/void cph:://' | grep -E "$pat"
