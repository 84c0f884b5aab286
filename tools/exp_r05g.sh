#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_ingest.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/exp_r05g.txt
python tools/ingest_probe.py 102400 2>&1 | grep -v amdgpu.ids >> gpurun_out/exp_r05g.txt
python tools/ingest_probe.py 102400 HGR 2>&1 | grep -v amdgpu.ids >> gpurun_out/exp_r05g.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 >> gpurun_out/exp_r05g.txt
cp ii-vision_amd/libiivision.so ab/libiiv_cur2.so
tools/ab_libs.sh "--steps 6 --warmup 1 --no-extras --no-emit --img --img-distinct 2048" ab/libiiv_base.so ab/libiiv_cur2.so ab/libiiv_base.so ab/libiiv_cur2.so >> gpurun_out/exp_r05g.txt 2>&1
cat gpurun_out/exp_r05g.txt
