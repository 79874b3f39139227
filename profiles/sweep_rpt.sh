for cfg in "2 1048576" "4 1048576" "2 1048576" "4 1048576" "4 4096"; do set -- $cfg
echo -n "rpt=$1 blocks=$2: "; timeout -k 10 120 python -u bench.py --steps 30 --warmup 5 --no-cpu-baseline --tune-rpt $1 --tune-blocks $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3), round(d['ms_per_step'],4))"
done
