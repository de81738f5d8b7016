# bench.py without the CPU legs: reads/s, kernel ms per pass, reads per tier
python bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-pairs 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms']), d['config']['tier_reads'])"
