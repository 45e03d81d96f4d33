V=arithmetic-circuits_amd/variants
echo "=== pytest ntt/qap/mgpu"; python -m pytest tests -x -q -m gpu -k "ntt or qap_h or mgpu or golden" 2>&1 | tail -4
ext() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'k2 %.2f' % d['roofline']['kernel_us'], 'ntt %.2f' % d['ntt']['us'], 'batch %.2f' % d['ntt']['batch']['us_per_transform'], 'qap_h %.1f' % d['qap_h']['us'], 'parity', d['ntt']['parity_vs_oracle'], d['qap_h']['parity_vs_oracle'])"; }
for rnd in 1 2; do
for v in base nttpre0 default; do
  if [ $v = default ]; then unset ACX_LIB; else export ACX_LIB=$V/libacx_$v.so; fi
  python bench.py --no-cpu --sustain 0 2>/dev/null | ext "bn254 $v"
done; done
for v in base nttpre0 default; do
  if [ $v = default ]; then unset ACX_LIB; else export ACX_LIB=$V/libacx_$v.so; fi
  python bench.py --no-cpu --sustain 0 --field bls12_381 2>/dev/null | ext "bls $v"
done
unset ACX_LIB
echo "=== mgpu W=1"; python bench.py --mgpu-devices 0 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us'], d['mgpu_qap_h'])"
echo "=== K2 counters"
for v in base hot lay1 pipe1; do
  ACX_LIB=$V/libacx_$v.so python tools/prof.py --out gpurun_out/prof_$v --groups sq,sq2,fetch,tcp1 --match sell -- python bench.py --no-cpu --no-ntt --sustain 0 --steps 20 2>&1 | tail -6
done
