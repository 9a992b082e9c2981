#!/bin/bash
set -x
mkdir -p gpurun_out/s11
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s11/pytest.log 2>&1
tail -5 gpurun_out/s11/pytest.log
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
for i in 1 2; do
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s11/k127_aos_$i.json 2>/dev/null
PPG_PATH_LAYOUT=soa timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s11/k127_soa_$i.json 2>/dev/null
done
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s11/k20_aos.json 2>/dev/null
PPG_PATH_LAYOUT=soa timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s11/k20_soa.json 2>/dev/null
timeout 300 python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s11/k127_aos_kernels.json 2>/dev/null
timeout 300 python bench.py --scene cbox --steps 255 --warmup 5 $Q > gpurun_out/s11/cbox_soa.json 2>/dev/null
PPG_PATH_LAYOUT=aos timeout 300 python bench.py --scene cbox --steps 255 --warmup 5 $Q > gpurun_out/s11/cbox_aos.json 2>/dev/null
timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 $Q > gpurun_out/s11/ship_aos.json 2>/dev/null
PPG_PATH_LAYOUT=soa timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 $Q > gpurun_out/s11/ship_soa.json 2>/dev/null
