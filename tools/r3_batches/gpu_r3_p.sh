#!/bin/bash
# round-3 GPU batch P: per-kernel times of the JPEG pixel stage, one lane per block vs eight lanes per block
cd "$(dirname "$0")/../.."
tools/profile_jpeg_kernels.sh > /dev/null 2>&1; cp gpurun_out/jpeg_kernels/kernels.txt gpurun_out/r3_p_kernels_bpl.txt
IFHIP_JPEG_IDCT8=1 tools/profile_jpeg_kernels.sh > /dev/null 2>&1; cp gpurun_out/jpeg_kernels/kernels.txt gpurun_out/r3_p_kernels_idct8.txt
echo BPL; grep -v forward gpurun_out/r3_p_kernels_bpl.txt; echo IDCT8; grep -v forward gpurun_out/r3_p_kernels_idct8.txt
