for cfg in "1.26 0" "1.6 4" "2.0 4" "2.0 0" "2.6 4"; do set -- $cfg; export AVIRHIP_GP_SOLO=$1; if [ $2 = 0 ]; then unset AVIRHIP_GP_MINCHUNK; else export AVIRHIP_GP_MINCHUNK=$2; fi
echo "solo $1 minchunk $2:"
for w in cfg1 nonint triple nonint_u8 up2_u8; do timeout 60 python bench.py --workload $w --no-cpu --no-sustained 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %-9s %.4f' % (d['workload'], d['ms_per_step']))"; done
timeout 100 python tools/bench_lancir.py 2>&1 | grep "path 0" | awk '{print "   " $2, $3, $4, $7, $8}'
done
