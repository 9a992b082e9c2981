#!/bin/bash
# GPU box: the wavefront kernels' grid by batch size (PPG_BLOCKS_SMALL workgroups for batches of at most PPG_SMALL_PATHS paths)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s27
timeout 300 python -m pytest tests -m gpu -x -q --timeout 120 -k "tuning or stragglers_records" 2>&1 | tail -3
V=("-|" "-|PPG_BLOCKS_SMALL=2048 PPG_SMALL_PATHS=4000000" "-|PPG_BLOCKS_SMALL=2048 PPG_SMALL_PATHS=16000000" "-|PPG_BLOCKS_SMALL=2048 PPG_SMALL_PATHS=32000000" "-|PPG_BLOCKS_SMALL=3072 PPG_SMALL_PATHS=16000000")
tools/ab.sh r06_s27/ab20 3 20 "${V[@]}"
tools/ab.sh r06_s27/ab127 2 127 "${V[@]}"
tools/ab.sh r06_s27/ab1023 1 1023 "${V[0]}" "${V[2]}" "${V[3]}"
