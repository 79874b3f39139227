for lds in 0 1 0 1; do
echo -n "program_in_lds=$lds: "; timeout -k 10 120 python -u bench.py --steps 30 --warmup 5 --no-cpu-baseline --tune-lds $lds 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"
done
