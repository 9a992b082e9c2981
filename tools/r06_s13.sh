#!/bin/bash
# GPU box: with the pair compaction in k_trace — triangles per BVH leaf, hand-over threshold of the tail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s13
V='-| -|PPG_BVH_LEAF=2 -|PPG_BVH_LEAF=4 -|PPG_BVH_LEAF=6 -|PPG_TAIL_MIN=1400000 -|PPG_TAIL_MIN=3000000'
tools/ab.sh r06_s13/ab20 2 20 $V
tools/ab.sh r06_s13/ab127 1 127 $V
