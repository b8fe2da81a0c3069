#!/bin/bash
# round 6: k_dnf with LDS producer / consumer counters instead of the per-step workgroup
# barrier -- parity (golden hashes in the GPU suite, fuzz_dnf against the reference),
# then same-box A/B against the barrier build (make dnfbar), three alternations
O=gpurun_out/r06/dnf_flags; mkdir -p $O
python tools/fuzz_dnf.py 400 91 2>&1 | tail -1 | tee $O/fuzz_dnf.txt
python -m pytest tests -m gpu -q -x -k "downsizing or dnf or golden or integer_ratio" 2>&1 | tail -2 | tee $O/pytest.txt
( for rep in 1 2 3; do LIBS="libavirhip_dnfbar.so libavirhip.so" bash tools/ab_env_libs.sh cfg4 half cfg4_u8 half_u8; done ) | tee $O/ab.txt
