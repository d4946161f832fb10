#!/bin/bash
# usage: tools/isa_fused.sh [mangled-substring]   -- compile the fused kernel, print resource usage and the order of memory / sync instructions
cd "$(dirname "$0")/../video-annotator_amd" && mkdir -p /tmp/isa
K=${1:-ILi8ELi0ELi0ELb0}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off $EXTRA -I../include -c csrc/vstab_warp_fused.hip -o /tmp/isa/fused.o --save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "k_warp_fused$K" | grep -E "VGPRs:|SGPRs:|Scratch|Occupancy" | sed 's/remark:.*:0://;s/\[-Rpass.*//' | tr '\n' ' '; echo
mv vstab_warp_fused-* /tmp/isa/ 2>/dev/null
awk -v k="k_warp_fused$K" '$0 ~ "^_ZN5vstab12"k"[A-Za-z0-9_]*:"{f=1} f{print} /s_endpgm/{if(f){exit}}' /tmp/isa/vstab_warp_fused-hip-amdgcn-amd-amdhsa-gfx950.s > /tmp/isa/kf.s
wc -l /tmp/isa/kf.s
