#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/exp_r05c.sh
bash tools/exp_r05d.sh
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/exp_r05e_tests.txt
cat gpurun_out/exp_r05e_tests.txt
