#!/bin/bash
# Print VGPR / scratch / occupancy per kernel for every .hip file (cross-compiles for gfx950, no GPU needed).
cd "$(dirname "$0")/../ao_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk -v file=$f '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /VGPRs:/ && !/Spill/ {v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /SGPRs:/ && !/Spill/ {sg=$0; sub(/.*SGPRs: /,"",sg); sub(/ \[.*/,"",sg)}
       /ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)}
       /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l)}
       /Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
       /LDS Size/ {cmd="echo " name " | c++filt"; cmd | getline d; close(cmd); sub(/\(.*/,"",d); printf "%-14s %-60s vgpr=%-4s sgpr=%-4s scratch=%-4s occ=%-2s lds=%s\n", file, d, v, sg, s, o, l}'
done
