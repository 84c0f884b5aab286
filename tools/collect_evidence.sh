#!/bin/bash
# After `gpurun -- 'bash tools/round_evidence_core.sh'`: copy what that call wrote under gpurun_out/ into profiles/ (tracked).
tag=${1:-r06}
set -e
cd "$(dirname "$0")/.."
cp gpurun_out/${tag}_pmc_latest.json profiles/pmc_latest.json
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
for m in dhgr hgr img; do
  sfx=$([ $m = dhgr ] && echo "" || echo "_$m")
  cp gpurun_out/${tag}_prof_$m/kernel_stats.csv profiles/${tag}_kernel_stats$sfx.csv
  cp gpurun_out/${tag}_prof_$m/pmc_summary.txt profiles/${tag}_pmc_summary$sfx.txt
  cp gpurun_out/${tag}_prof_$m/bench_under_rocprof.json profiles/${tag}_bench_under_rocprof$sfx.json
done
for f in coherent img static config5 config3 config3img fourth; do cp gpurun_out/${tag}_bench_$f.json profiles/; done
cp gpurun_out/${tag}_rehearse_2ranks_one_gpu.json profiles/
echo "profiles/ <- gpurun_out/${tag}_* (build $(python3 -c "import json; print(json.load(open('profiles/${tag}_bench.json'))['build_id'])"))"
