#!/bin/bash
# The measurements a round's profiles/ files come from, in one gpurun call:
#   gpurun --timeout 3000 -- 'bash tools/round_evidence.sh r05'
# Writes gpurun_out/<tag>_*; copy what is to be kept into profiles/.
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
for v in "--coherent" "--img" "--static" "--palette IIGS" "--mode HGR" "--mode HGR --img" "--fourth"; do
  n=$(echo "$v" | tr -d ' -')
  python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline $v > $out/${tag}_bench_$n.json 2>/dev/null
done
python tools/profile_summary.py $out/${tag}_prof_dhgr --steps 2 --warmup 1 --no-extras --no-cpu-baseline --greedy shared > /dev/null 2>&1
python tools/profile_summary.py $out/${tag}_prof_hgr --mode HGR --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
python tools/profile_summary.py $out/${tag}_prof_img --img --img-distinct 2048 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
(python tools/ingest_probe.py 102400; python tools/ingest_probe.py 102400 HGR) 2>&1 | grep -v amdgpu.ids > $out/${tag}_ingest_probe.txt
python tests/fuzz_parity.py ${FUZZ_ROUNDS:-500} 2>&1 | tail -3 > $out/${tag}_fuzz_parity.txt
(python tests/long_parity.py 1000 8 wave; python tests/long_parity.py 1000 8 wave 5 img) 2>&1 | tail -12 > $out/${tag}_long_parity.txt
