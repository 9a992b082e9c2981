#!/bin/bash
# round 5, session 5: the whole GPU suite on the current build; the reference-log probe (tools/ref_log_probe.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
cd /tmp; timeout 600 python $R/tools/ref_log_probe.py 63 2>&1 | tail -40
