#!/bin/bash
# converging content (S-static) against S-iid, per emitted opcode: round 2's library and this one
for args in "" "--static" "--static --repeat 4"; do
  echo "== input: ${args:-S-iid}"
  tools/ab_libs.sh "--steps 4 --warmup 2 --no-extras --streams 14336 $args" ab/libiiv_r02.so ab/libiiv_base.so
done
