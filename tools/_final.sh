echo "=== gpu suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -5
echo "=== bench"; timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_line.json; python -c "
import json; d=json.load(open('gpurun_out/bench_line.json')); print(d['value'], d['roofline']['kernel_us'], d['roofline']['frac'], d['roofline'].get('frac_traffic'), d['parity_vs_oracle'], d['ntt']['us'], d['qap_h']['us'], d['r1cs_small_coeff']['us_per_launch'], d['cpu_baseline']['value'])"
