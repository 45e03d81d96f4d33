V=arithmetic-circuits_amd/variants
echo "=== fuzz hotp3"; ACX_LIB=$V/libacx_hotp3.so timeout 300 python tools/fuzz_r1cs.py 6 2>&1 | tail -2
timeout 600 python tools/k2_ab.py $V/libacx_base.so $V/libacx_hotp3.so $V/libacx_hotp6.so 2>&1 | tail -4
echo "=== dist tests with store-order tables"; timeout 900 python -m pytest tests -x -q -m gpu -k "mgpu or dist or rccl or force_dist" 2>&1 | tail -4
echo "=== dist budget"; timeout 600 python tools/dist_budget.py 2>&1 | tail -16
