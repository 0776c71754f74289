#!/bin/bash
# compile mrx_stream_bits.hip alone and print the register / scratch / occupancy figures of its kernels
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage -c /root/repo/mojo_regex_amd/csrc/mrx_stream_bits.hip -o /tmp/co/sb.o 2>&1 | grep -E "error|warning|Function Name|VGPRs:|SGPRs:|Occupancy|Scratch" | paste - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/mrx_stream_bits.hip:[0-9]*:1: remark://g; s/_ZN3mrx12_GLOBAL__N_1//'
