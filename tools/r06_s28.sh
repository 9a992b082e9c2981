#!/bin/bash
# GPU box: grid by batch size as shipped (2048 workgroups up to 13 M paths) against one grid for all
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s28
tools/ab.sh r06_s28/ab20 2 20 "-|PPG_BLOCKS_SMALL=0" "-|"
tools/ab.sh r06_s28/ab127 1 127 "-|PPG_BLOCKS_SMALL=0" "-|"
tools/ab.sh r06_s28/ab1023 2 1023 "-|PPG_BLOCKS_SMALL=0" "-|"
