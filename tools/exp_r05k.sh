#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_ingest.py tests/test_gpu_transcode_tool.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/exp_r05k.txt
python tools/ingest_probe.py 102400 2>&1 | grep -v amdgpu.ids >> gpurun_out/exp_r05k.txt
python tools/ingest_probe.py 102400 HGR 2>&1 | grep -v amdgpu.ids >> gpurun_out/exp_r05k.txt
cat gpurun_out/exp_r05k.txt
