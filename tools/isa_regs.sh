#!/bin/bash
# register / scratch use of every kernel in a .hip file (cross-compiles, no GPU needed)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Iinclude -Itracer_amd/csrc --cuda-device-only -S -o /tmp/isa/k.s "$1" 2>/dev/null
awk '/^  - \.agpr_count/{a=$3} /\.name:/{n=$2} /\.private_segment_fixed_size:/{p=$2} /\.sgpr_count:/{s=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{print n, "vgpr", v, "sgpr", s, "scratch", p, "spill", $2}' /tmp/isa/k.s
