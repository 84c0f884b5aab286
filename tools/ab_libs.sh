#!/bin/bash
# A/B: run bench.py against several builds of the library (IIV_LIB override) on one box.
#   tools/ab_libs.sh "bench args" lib1.so lib2.so ...
args="$1"; shift
for lib in "$@"; do
  IIV_LIB=$PWD/$lib python bench.py --no-cpu-baseline $args 2>/tmp/ab_libs.err | tail -1 | python -c "
import sys, json
s = sys.stdin.read()
try:
    j = json.loads(s)
    print('%-32s %10.0f fps  greedy %.4f ms  prologue %.4f ms' % ('$lib', j['value'], j['roofline']['avg_launch_ms'], j['roofline_prologue']['avg_launch_ms']))
except Exception:
    print('%-32s FAILED; stderr tail:' % '$lib')
    print(''.join(open('/tmp/ab_libs.err').readlines()[-6:]))"
done
