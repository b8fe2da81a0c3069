#!/bin/bash
# k_gv chunk height x lane width sweep on workload $1 (same box, one run)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for wp in ${WPS:-1 2}; do for c in ${CHUNKS:-0 10 14 20 28 35 47 56 70 94}; do
  export AVIRHIP_GV_WP=$wp
  if [ $c != 0 ]; then export AVIRHIP_GV_CHUNK=$c; else unset AVIRHIP_GV_CHUNK; fi
  echo -n "wp $wp chunk $c: "
  bash tools/prof_gpass.sh ${1:-nonint} 2>&1 | grep "k_gv" | sed 's/.*Params)",//' | cut -d, -f1-3
done; done
