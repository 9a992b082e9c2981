#!/bin/bash
# GPU box: k_shade<rest> on the second stream beside k_shade<common> — parity, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s25
timeout 400 python -m pytest tests -m gpu -x -q --timeout 120 -k "golden or room or kitchen or stragglers or spaceship or tuning or full_size or two_ranks or material or coincident" 2>&1 | tail -4
tail -1 /dev/null; rocm-smi --showuse 2>/dev/null | head -8; tools/ab.sh r06_s25/ab20 3 20 "-|PPG_NO_REST_BESIDE=1" "-|"
tools/ab.sh r06_s25/ab127 2 127 "-|PPG_NO_REST_BESIDE=1" "-|"
tools/ab.sh r06_s25/ab1023 1 1023 "-|PPG_NO_REST_BESIDE=1" "-|"
