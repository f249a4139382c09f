#!/bin/bash
# scratch/isa_build.sh <file stem in csrc>: compile one source with resource remarks, keep the ISA under scratch/tmp
mkdir -p /root/repo/scratch/tmp && cd /root/repo/scratch/tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include -c /root/repo/tilingnn_amd/csrc/$1.hip -o $1_chk.o -Rpass-analysis=kernel-resource-usage -save-temps=obj 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|LDS Size" 
