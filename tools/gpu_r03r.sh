#!/bin/bash
# GPU box: lane utilisation of the persistent trace kernels in batch mode (32 frames per launch sequence, like the bench) from a -DPT_HIST build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O
PT_LIB=$PWD/vk_raytrace_amd/variants/libptmi_hist.so timeout 300 python tools/gpu_hist.py 32 > $O/hist_batch32.txt 2>&1
cat $O/hist_batch32.txt
