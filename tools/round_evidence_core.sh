#!/bin/bash
# The build-id dependent part of tools/round_evidence.sh alone (counter runs, then the bench lines that quote them):
#   gpurun --timeout 2400 -- "bash tools/round_evidence_core.sh"
tag=r06
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
(IIV_BENCH_REHEARSE_ON_ONE_GPU=1 python bench.py --gpus 2 --streams 3584 --steps 4 --warmup 1 --no-extras --no-cpu-baseline --config 5) > $out/${tag}_rehearse_2ranks_one_gpu.json 2>/dev/null
