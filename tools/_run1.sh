python -m pytest tests/test_hip_parity.py -q -x -s -k "rdn_trunk_halo_resident or tile_batch_is_bitwise or (full_c3_tile_vs_reference and fp32)" 2>&1 | grep "rdn trunk\|passed\|failed\|Error\|assert" | head
python3 tools/kernel_lab.py --quick fp32=fp32 2>&1 | tail -1
python3 bench.py --workload c3 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 fp32', d['ms_per_step'], {k:round(v,1) for k,v in list(d['kernels_ms_per_step'].items())[:4]})"
