#!/bin/bash
# register / spill counts of the 16384-point fused kernels (plain, pipelined, accumulating): tools/k7_regs.sh [extra hipcc flags]
cd $(dirname $0)/../libsdr_amd/csrc; mkdir -p _asm
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I. "$@" -S --cuda-device-only -o _asm/fftconv.s fftconv.hip 2>&1 | grep -v hip-link | head
for k in ILi14ELb0ELi1024ELb0ELi0ELi0E ILi14ELb0ELi1024ELb0ELi4ELi2E ILi14ELb0ELi1024ELb0ELi4ELi0E ILi14ELb0ELi1024ELb0ELi4ELi4E ILi14ELb0ELi1024ELb0ELi2ELi0E; do echo $k $(grep -A40 "\.name:.*fftconv_fused_kernel$k" _asm/fftconv.s | grep -E "vgpr_count|vgpr_spill|private_segment" | tr -s ' ' | tr '\n' ' '); done
