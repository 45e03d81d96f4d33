echo "=== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
echo "=== gpu suite"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -5
echo "=== profile"; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 1500 python tools/prof.py --out gpurun_out/prof_r03 --groups sq,fetch,write --pass-timeout 400 -- python bench.py --no-cpu 2>&1 | tail -40
