#!/bin/bash
out=gpurun_out
mkdir -p $out
timeout 600 tools/issue_probe > $out/r06_issue_probe.txt 2>&1
for rep in 1 2; do tools/ab_libs.sh "--mode HGR --steps 4 --warmup 2 --no-extras" ab/libiiv_base.so ab/libiiv_hgrmtreg.so; done 2>&1 | tee $out/r06_step2_ab.txt
(IIV_LIB=$PWD/ii-vision_amd/libiivision_stamps.so python tools/prologue_stamps.py HGR; IIV_LIB=$PWD/ii-vision_amd/libiivision_stamps.so python tools/prologue_stamps.py) 2>&1 | grep -v amdgpu.ids | tee $out/r06_step2_stamps.txt
cat $out/r06_issue_probe.txt
