#!/bin/bash
# round-2 GPU session 3: A/B experiments on the KITCHEN workload (queue sort by BSDF type, tail threshold, BSDF mix cost)
set -x
mkdir -p gpurun_out/s3
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_textures.py tests/test_gpu_parity.py -m gpu -x -q -k "textures or glossy or rough or golden or stepwise or null or spheres" > gpurun_out/s3/pytest.log 2>&1; tail -3 gpurun_out/s3/pytest.log
B="python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary"
timeout 300 $B > gpurun_out/s3/sort_on.json 2>gpurun_out/s3/sort_on.err; tail -c 1800 gpurun_out/s3/sort_on.json
PPG_NO_SORT=1 timeout 300 $B > gpurun_out/s3/sort_off.json 2>/dev/null; tail -c 1800 gpurun_out/s3/sort_off.json
PPG_TAIL_THRESHOLD=32768 timeout 300 $B --no-roofline > gpurun_out/s3/tail32k.json 2>/dev/null; tail -c 500 gpurun_out/s3/tail32k.json
PPG_TAIL_THRESHOLD=4000000 timeout 300 $B --no-roofline > gpurun_out/s3/tail4m.json 2>/dev/null; tail -c 500 gpurun_out/s3/tail4m.json
timeout 300 $B --all-diffuse > gpurun_out/s3/all_diffuse.json 2>/dev/null; tail -c 1800 gpurun_out/s3/all_diffuse.json
PPG_BLOCKS=4096 timeout 300 $B --no-roofline > gpurun_out/s3/blocks4096.json 2>/dev/null; tail -c 500 gpurun_out/s3/blocks4096.json
PPG_BLOCKS=1024 timeout 300 $B --no-roofline > gpurun_out/s3/blocks1024.json 2>/dev/null; tail -c 500 gpurun_out/s3/blocks1024.json
