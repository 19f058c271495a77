#!/bin/sh
# usage: resusage.sh file.hip [extra flags] -> kernel, VGPRs, SGPRs, scratch bytes/lane, occupancy (compile-time view)
f="$1"; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "error|Function Name|TotalSGPRs|VGPRs:|ScratchSize|Occupancy" \
 | sed -E 's/.*remark: *//; s/\[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n,v,s,sc,o; n=$3} /VGPRs:/{v="v="$2} /TotalSGPRs/{s="s="$2} /ScratchSize/{sc="scr="$3} /Occupancy/{o="occ="$3} END{print n,v,s,sc,o}' | c++filt | sed -E 's/ipa:://g; s/\(.*\)//'
