python -m pytest tests/test_hip_parity.py -q -x -k "tile_streams_are_bitwise" 2>&1 | tail -2
for p in f16 fp32; do for e in "" "--encoder-ahead"; do
python3 bench.py --workload c3 --precision $p $e --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 $p $e', d['ms_per_step'])"
done; done
