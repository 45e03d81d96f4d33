V=arithmetic-circuits_amd/variants
python tools/k2_ab.py $V/libacx_base.so $V/libacx_pipe1.so $V/libacx_pipe1w5.so $V/libacx_hot.so $V/libacx_lay1.so $V/libacx_lay1w6.so $V/libacx_hotlay1w6.so 2>&1 | tail -12
for v in hot lay1w6; do echo "fuzz $v"; ACX_LIB=$V/libacx_$v.so python tools/fuzz_r1cs.py 8 2>&1 | tail -2; done
echo "=== mgpu W=1"; python bench.py --mgpu-devices 0 --no-cpu 2>&1 | tail -1
echo "=== mgpu W=2"; python bench.py --mgpu-devices 0,0 --copies 16 --no-cpu 2>&1 | tail -1
echo "=== mgpu W=8"; python bench.py --mgpu-devices 0,0,0,0,0,0,0,0 --copies 4 --no-cpu 2>&1 | tail -1
