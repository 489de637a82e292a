#!/bin/bash
mkdir -p gpurun_out/r5
CODEPS_HALO_N16=1 python tools/conv_layers.py 2>/dev/null | grep -E "16 k3|->16 k3|halo3n16|32\+0->16|16\+0->32" > gpurun_out/r5/n16_on.txt
CODEPS_HALO_N16=0 python tools/conv_layers.py 2>/dev/null | grep -E "16 k3|->16 k3|32\+0->16|16\+0->32" > gpurun_out/r5/n16_off.txt
echo ON; cat gpurun_out/r5/n16_on.txt | cut -c1-140
echo OFF; cat gpurun_out/r5/n16_off.txt | cut -c1-140
