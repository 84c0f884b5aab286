#!/bin/bash
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tables.py tests/test_gpu_encode.py tests/test_gpu_configs.py tests/test_gpu_properties.py -x -q -m gpu > $out/r06_step3_tests.txt 2>&1
tail -3 $out/r06_step3_tests.txt
for rep in 1 2; do tools/ab_libs.sh "--mode HGR --steps 4 --warmup 2 --no-extras" ab/libiiv_base.so ab/libiiv_hgrmtreg.so; done 2>&1 | tee $out/r06_step3_ab.txt
bash tools/pmc_quick.sh --mode HGR --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | tee $out/r06_step3_pmc_hgr.txt
