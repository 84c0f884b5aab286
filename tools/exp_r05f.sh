#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== greedy A/B: round-4 kernel (ab/libiiv_base.so) vs current (one-pass top-2, xad tie test, lean apply)"
cp ii-vision_amd/libiivision.so ab/libiiv_cur.so
for a in "" "--img --img-distinct 2048" "--mode HGR"; do
  echo "-- args: $a"
  tools/ab_libs.sh "--steps 6 --warmup 1 --no-extras --no-emit $a" ab/libiiv_base.so ab/libiiv_cur.so ab/libiiv_base.so ab/libiiv_cur.so
done
echo "== diffusion kernel residency (LDS pad per 4-wave block)"
for pad in 0 8192 16384 32768 65536; do
  echo "-- pad $pad"; IIV_EXP_DIFF_LDS_PAD=$pad python tools/ingest_probe.py 102400 2>&1 | grep -v amdgpu.ids | tail -1
done
python tools/ingest_probe.py 102400 2>&1 | grep -v amdgpu.ids
python tools/ingest_probe.py 102400 HGR 2>&1 | grep -v amdgpu.ids
} > gpurun_out/exp_r05f.txt 2>&1
cat gpurun_out/exp_r05f.txt
