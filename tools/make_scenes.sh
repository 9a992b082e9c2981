#!/bin/bash
# Regenerates the flat scene files the benchmark and tests/test_real_scenes.py read (scratch/*.ppgs, untracked: they are conversions of the
# reference's scene DATA) from the reference checkout, and checks them against scratch/SHA256SUMS (tracked).  Development container only
# (/root/reference does not exist on the GPU box; the files travel with the snapshot).
#   tools/make_scenes.sh            convert, then verify
#   tools/make_scenes.sh --check    verify only
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REF=${PPG_REFERENCE:-/root/reference}
cd "$R"
if [ "$1" != "--check" ]; then
  mkdir -p scratch
  export PYTHONPATH="$R/practical-path-guiding_amd"
  # BASELINE.json configs[2]: KITCHEN, improved preset, at the benchmark's 1280x720 (283 of 289 meshes: six are missing from the checkout)
  python -m ppg_host "$REF/scenes/kitchen/kitchen-improved.xml" --lenient --data-dir "$REF/mitsuba/data" --size 1280x720 --ppgs scratch/kitchen-improved.ppgs
  # BASELINE.json configs[3] geometry: SPACESHIP (84 of 86 meshes)
  python -m ppg_host "$REF/scenes/spaceship/spaceship.xml" --lenient --data-dir "$REF/mitsuba/data" --ppgs scratch/spaceship.ppgs
fi
(cd scratch && sha256sum -c SHA256SUMS)
