for d in 0 7 23; do
CIAOSR_HIP_LIB=$PWD/ciaosr_amd/csrc/variants/libdbg.so CIAOSR_DENSE_DBG=$d python3 bench.py --workload c3s --precision f16 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3s f16 DBG=$d', d['ms_per_step'], d['kernels_ms_per_step'].get('enc_dense_f16'))"
done
