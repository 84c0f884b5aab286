#!/bin/bash
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tables.py tests/test_gpu_encode.py -x -q -m gpu > $out/r06_step4_tests.txt 2>&1
tail -3 $out/r06_step4_tests.txt
for rep in 1 2; do tools/ab_libs.sh "--mode HGR --steps 4 --warmup 2 --no-extras" ab/libiiv_base.so ab/libiiv_hgrfe.so ab/libiiv_hgrmtlds.so; done 2>&1 | tee $out/r06_step4_ab.txt
bash tools/pmc_quick.sh --mode HGR --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | grep -A 17 prologue | tee $out/r06_step4_pmc_hgr.txt
