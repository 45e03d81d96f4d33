echo "=== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | grep -E "smoke|Error|error" | tail -3
echo "=== gpu suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -8
echo "=== pass probe"; timeout 600 python tools/pass_probe.py 2>&1 | grep LP
