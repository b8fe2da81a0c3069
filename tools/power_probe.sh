#!/bin/bash
cd $GRAFT_REPO_ROOT
(timeout 80 python bench.py --no-cpu --steps 200000 --warmup 10 > /tmp/b.json 2>/dev/null) &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' '; echo; sleep 1; done
wait $BP
tail -1 /tmp/b.json | cut -c1-200
rocm-smi --showmaxpower 2>/dev/null | grep -i power
