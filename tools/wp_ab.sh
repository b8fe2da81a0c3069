#!/bin/bash
# parity subset + in-run A/B of the pass kernels: libraries $LIBS (default: the
# built one and, if present, lib/libavirhip_head.so = the previous commit's
# gpass.hip) x lane widths $WPS. Boxes differ by up to 1.4x in clocks: only
# numbers from the same run compare.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "pass_kernels or band_execution or random_geom or lancir" 2>&1 | tail -3
for lib in ${LIBS:-libavirhip_head.so libavirhip.so}; do
  [ -f avir_amd/lib/$lib ] || continue
  export AVIRHIP_LIB=$PWD/avir_amd/lib/$lib
  for wp in ${WPS:-2}; do
    export AVIRHIP_GV_WP=$wp
    echo "== $lib WP=$wp"
    for W in ${WLS:-nonint}; do
      bash tools/prof_gpass.sh $W 2>&1 | grep "k_g[vh]" | sed 's/.*Params)",//' | cut -d, -f1-3
    done
    python tools/bench_lancir.py 2>&1 | tail -8 | grep "path 0"
  done
done
