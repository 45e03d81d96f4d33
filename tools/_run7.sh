echo "=== full gpu suite"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5
echo "=== dist budget"; timeout 600 python tools/dist_budget.py 2>&1 | tail -16
echo "=== bench"; timeout 600 python bench.py 2>&1 | tail -1
