#!/bin/bash
# register / spill table of one kernel source:  tools/r04/isa.sh conv_rw [extra hipcc flags]
F=$1; shift
cd /root/repo/cbim-medical-image-segmentation_amd/csrc
mkdir -p /tmp/isa
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wno-unused-result -S --cuda-device-only "$@" $F.hip -o /tmp/isa/$F.s 2>&1 | grep -v "warning: argument unused"
grep "\.vgpr_count\|vgpr_spill\|\.name:\|sgpr_spill" /tmp/isa/$F.s | sed 's/.*\.name: *//' | paste - - - - | grep -v warm | awk '{printf "%-70s sgpr_spill %3s vgpr %3s vgpr_spill %3s\n", $1, $3, $5, $7}'
