for v in "1 2" "2 2" "2 3" "2 4"; do set -- $v
CIAOSR_DENSE_V=$1 CIAOSR_DENSE_SLICES=$2 python3 bench.py --workload c3 --precision f16 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 f16 V=$1 slices=$2', d['ms_per_step'], {k:v for k,v in list(d['kernels_ms_per_step'].items())[:4]})"
done
