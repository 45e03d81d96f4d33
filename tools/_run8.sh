echo "=== mgpu tests"; timeout 900 python -m pytest tests/test_mgpu.py tests/test_gpu_parity.py -x -q -m gpu -k "mgpu or json or plain_c or dist" 2>&1 | grep -E "passed|failed|rror" | tail -5
for dv in 0 0,0; do
echo "=== mgpu bench devices $dv"; timeout 600 python bench.py --mgpu-devices $dv --copies $([ $dv = 0 ] && echo 32 || echo 16) --no-cpu 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us'], d['mgpu_qap_h']['us'], d['mgpu_qap_h']['accepts_valid_rejects_corrupt'], d['parity_vs_oracle'])"
done
