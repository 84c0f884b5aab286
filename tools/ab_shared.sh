for g in plain shared; do for S in 12288 14336 16384; do
python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --greedy $g --streams $S 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$g', $S, round(d['value']), 'greedy ms', round(d['roofline']['avg_launch_ms'],4), 'prologue ms', round(d['roofline_prologue']['avg_launch_ms'],4))"
done; done
