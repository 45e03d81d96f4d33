V=arithmetic-circuits_amd/variants
ext() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ntt %.2f' % d['ntt']['us'], 'batch %.2f' % d['ntt']['batch']['us_per_transform'], 'qap_h %.1f' % d['qap_h']['us'], 'parity', d['ntt']['parity_vs_oracle'], d['qap_h']['parity_vs_oracle'])"; }
for rnd in 1 2; do
for v in base ntt_pre0_dflt ntt_pre1_dflt ntt_pre1_maxocc ntt_pre1_minreg; do
  ACX_LIB=$V/libacx_$v.so python bench.py --no-cpu --sustain 0 2>/dev/null | ext "bn254 $v"
done; done
echo "=== new tests"; python -m pytest tests -x -q -m gpu -k "json or plain_c" 2>&1 | tail -3
