#!/bin/bash
# occupancy trend of the LDS-shared greedy kernel: W streams per workgroup, 2 workgroups per CU
run() { tools/ab_libs.sh "--steps 4 --warmup 1 --no-extras --greedy shared --streams $2" ab/libiiv_$1.so | sed "s/^/S=$2 /"; }
run r8 12288; run r9 13824; run r10 15360
