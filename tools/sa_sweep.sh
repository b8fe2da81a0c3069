mkdir -p gpurun_out/r4s; rm -f gpurun_out/r4s/s.txt
W=${1:-readme}; HA=${2:-52}; VA=${3:-19}
echo "== $W: auto" >> gpurun_out/r4s/s.txt
timeout 100 python bench.py --no-cpu --workload $W --steps 200 --warmup 10 2>&1 | tail -1 | cut -c1-90 >> gpurun_out/r4s/s.txt
for h in 20 26 32 40 48 52 60 70 80 96 120 160; do
echo -n "H=$h V=$VA " >> gpurun_out/r4s/s.txt
AVIRHIP_SA_CHUNK=$h AVIRHIP_SA_CHUNK_V=$VA timeout 100 python bench.py --no-cpu --workload $W --steps 200 --warmup 10 2>&1 | tail -1 | cut -c30-70 >> gpurun_out/r4s/s.txt
done
for v in 8 10 12 14 16 19 22 26 32 40 54 64; do
echo -n "H=auto V=$v " >> gpurun_out/r4s/s.txt
AVIRHIP_SA_CHUNK_V=$v timeout 100 python bench.py --no-cpu --workload $W --steps 200 --warmup 10 2>&1 | tail -1 | cut -c30-70 >> gpurun_out/r4s/s.txt
done
cat gpurun_out/r4s/s.txt
