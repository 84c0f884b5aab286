#!/bin/bash
# round-5 experiment A: occupancy ceiling of the LDS-shared form (timing only: aliased LDS)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== base vs aliased variants (shared form forced), S-iid 14336"
tools/ab_libs.sh "--steps 4 --warmup 1 --no-extras --no-emit --greedy shared" ab/libiiv_base.so ab/libiiv_w8alias.so ab/libiiv_w10alias.so ab/libiiv_w12alias.so
echo "== same on S-img"
tools/ab_libs.sh "--steps 4 --warmup 1 --no-extras --no-emit --greedy shared --img --img-distinct 2048" ab/libiiv_base.so ab/libiiv_w12alias.so
} > gpurun_out/exp_r05a.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/exp_r05a_tests.txt
cat gpurun_out/exp_r05a.txt gpurun_out/exp_r05a_tests.txt
