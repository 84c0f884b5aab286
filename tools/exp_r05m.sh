#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/exp_r05m.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/exp_r05m.txt
cat gpurun_out/exp_r05m.txt
