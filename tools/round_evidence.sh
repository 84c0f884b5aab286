#!/bin/bash
# The measurements a round's profiles/ files come from, in one gpurun call:
#   gpurun --timeout 3600 -- 'bash tools/round_evidence.sh r06'
# Writes gpurun_out/<tag>_*; copy what is to be kept into profiles/ (gpurun_out/<tag>_pmc_latest.json -> profiles/pmc_latest.json).
# Order matters: the counter runs come FIRST and their summary replaces profiles/pmc_latest.json on the box, so that the
# bench lines behind them quote counters of this very build (bench.py refuses any other: build_id).
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export IIV_PMC_LATEST_OUT=$PWD/$out/${tag}_pmc_latest.json
rm -f $IIV_PMC_LATEST_OUT
python tools/profile_summary.py $out/${tag}_prof_dhgr --steps 2 --warmup 1 --no-extras --no-cpu-baseline --greedy shared > /dev/null 2>&1
python tools/profile_summary.py $out/${tag}_prof_hgr --mode HGR --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
python tools/profile_summary.py $out/${tag}_prof_img --img --img-distinct 2048 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
if [ -s $IIV_PMC_LATEST_OUT ]; then cp $IIV_PMC_LATEST_OUT profiles/pmc_latest.json; fi
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
for v in "--coherent" "--img" "--static" "--config 5" "--config 3" "--config 3 --img" "--fourth"; do
  n=$(echo "$v" | tr -d ' -')
  python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline $v > $out/${tag}_bench_$n.json 2>/dev/null
done
tools/issue_probe > $out/${tag}_issue_probe.txt 2>&1
(python tools/ingest_probe.py 102400; python tools/ingest_probe.py 102400 HGR) 2>&1 | grep -v amdgpu.ids > $out/${tag}_ingest_probe.txt
bash tools/pmc_cmd.sh diffusion tools/ingest_probe.py 51200 > $out/${tag}_ingest_diffusion_pmc.txt 2>&1
python tests/fuzz_parity.py ${FUZZ_ROUNDS:-500} 2>&1 | tail -3 > $out/${tag}_fuzz_parity.txt
(python tests/long_parity.py 1000 8 wave; python tests/long_parity.py 1000 8 wave 5 img) 2>&1 | tail -12 > $out/${tag}_long_parity.txt
(IIV_BENCH_REHEARSE_ON_ONE_GPU=1 python bench.py --gpus 2 --streams 3584 --steps 4 --warmup 1 --no-extras --no-cpu-baseline --config 5) > $out/${tag}_rehearse_2ranks_one_gpu.json 2>/dev/null
(python -m pytest tests -x -q -m gpu 2>&1 | tail -4; python __graft_entry__.py smoke 2>&1 | tail -1) > $out/${tag}_gpu_tests_final.txt
