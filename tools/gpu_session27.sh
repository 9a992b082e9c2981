#!/bin/bash
set -x
mkdir -p gpurun_out/s27
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "tuning_switches" > gpurun_out/s27/pytest.log 2>&1
tail -15 gpurun_out/s27/pytest.log
