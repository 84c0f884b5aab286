#!/bin/bash
out=gpurun_out
mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_encode.py tests/test_gpu_configs.py tests/test_gpu_properties.py tests/test_gpu_fourth.py -x -q -m gpu > $out/r06_step5_tests.txt 2>&1
tail -15 $out/r06_step5_tests.txt
for rep in 1 2; do tools/ab_libs.sh "--mode HGR --steps 4 --warmup 2 --no-extras" ab/libiiv_hgr1half.so ab/libiiv_base.so; done 2>&1 | tee $out/r06_step5_ab.txt
