#!/bin/bash
# round-5 experiment D: launch tail vs batch size (persistent workgroups: 4096 slots at W = 8, 6144 at W = 12)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for S in 8192 12288 14336; do
  echo "== S=$S"
  tools/ab_libs.sh "--steps 4 --warmup 1 --no-extras --no-emit --greedy shared --streams $S" ab/libiiv_base.so ab/libiiv_w12alias.so
done
} > gpurun_out/exp_r05d.txt 2>&1
cat gpurun_out/exp_r05d.txt
