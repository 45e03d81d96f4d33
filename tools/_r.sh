V=arithmetic-circuits_amd/variants
timeout 600 python tools/k2_ab.py $V/libacx_base.so $V/libacx_cpre.so 2>&1 | tail -2
ACX_LIB=$V/libacx_trace.so timeout 300 python tools/k2_trace.py 2>&1 | tail -13
ACX_LIB=$V/libacx_cpre.so timeout 300 python tools/fuzz_r1cs.py 8 2>&1 | tail -1
