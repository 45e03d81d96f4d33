echo "=== gpu suite"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -5
echo "=== dist budget"; timeout 600 python tools/dist_budget.py 2>&1 | tail -17
echo "=== force-dist bench"; timeout 900 python bench.py --force-dist --no-cpu --no-ntt 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['dist_ntt']['us'], d['dist_ntt']['parity_vs_oracle'], d['dist_qap_h']['us'], d['dist_qap_h']['accepts_valid_rejects_corrupt'])"
