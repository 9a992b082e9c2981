#!/bin/bash
# GPU box, round 3, session 16: box-to-box variance (clocks), A/B of the cooperative traversal and the cooperative prefetch on ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s16
mkdir -p $OUT
cd $R && timeout 600 python -m pytest tests/test_real_scenes.py tests/test_gpu_parity.py -m gpu -x -q -k "real or room or tuning or kitchen or null or envmap" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rocm-smi --showclocks --showperflevel --showpower --showtemp > $OUT/smi_before.txt 2>&1
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
L=$R/practical-path-guiding_amd/lib
for rep in 1 2 3 4; do
  $B > $OUT/pf_$rep.json 2>> $OUT/err.log
  PPG_HIP_LIB=$L/libppg_hip_nopf.so $B > $OUT/nopf_$rep.json 2>> $OUT/err.log
  PPG_HIP_LIB=$L/libppg_hip_nocoop.so $B > $OUT/nocoop_$rep.json 2>> $OUT/err.log
done
rocm-smi --showclocks --showperflevel --showpower --showtemp > $OUT/smi_mid.txt 2>&1
# does a longer warm-up change the timed 20 passes (clock ramp)?
python $R/bench.py --steps 20 --warmup 100 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/pf_warm100.json 2>> $OUT/err.log
python $R/bench.py --steps 20 --warmup 100 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/pf_warm100b.json 2>> $OUT/err.log
for v in pf nopf nocoop; do
  l=$L/libppg_hip_$v.so; [ $v = pf ] && l=$L/libppg_hip.so
  PPG_HIP_LIB=$l python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/${v}_127.json 2>> $OUT/err.log
  PPG_HIP_LIB=$l python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call > $OUT/${v}_timing.json 2>> $OUT/err.log
done
# clocks sampled while a long render runs
( for i in $(seq 1 12); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT/smi_during.txt &
python $R/bench.py --steps 511 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/pf_511.json 2>> $OUT/err.log
wait
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s16.//'
python - <<P
import json
for v in ('pf','nopf','nocoop'):
    d=json.load(open('$OUT/%s_timing.json'%v)); print(v, d['roofline']['kernels_ms'].get('k_tail'), d['roofline'].get('tail_critical_path'))
P
cat $OUT/smi_before.txt | grep -E "clk|Perf|Power" | head -12
