#!/bin/bash
# GPU box: no margin bounces for batches below the hand-over threshold — parity subset, A/B against the previous build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s31
timeout 400 python -m pytest tests -m gpu -x -q --timeout 120 -k "tuning or stragglers or golden or kitchen or two_ranks" 2>&1 | tail -3
tools/ab.sh r06_s31/ab20 3 20 "libppg_hip_prev.so|" "-|"
