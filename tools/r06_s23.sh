#!/bin/bash
# GPU box: persistent workgroups of the path kernels (PPG_BLOCKS, 4096) with k_trace at eight workgroups a CU
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s23
tools/ab.sh r06_s23/ab20 2 20 "-|" "-|PPG_BLOCKS=2048" "-|PPG_BLOCKS=3072" "-|PPG_BLOCKS=6144" "-|PPG_BLOCKS=8192"
tools/ab.sh r06_s23/ab127 1 127 "-|" "-|PPG_BLOCKS=2048" "-|PPG_BLOCKS=3072" "-|PPG_BLOCKS=6144" "-|PPG_BLOCKS=8192"
