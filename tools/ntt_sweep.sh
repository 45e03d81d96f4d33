# development sweep of digit splits (ACX_NTT_DIGITS) on an MI355X:  bash tools/ntt_sweep.sh "20:8+12 20:6+6+8" ...
for spec in "$@"; do
  for s in $spec; do
    ln=${s%%:*}; d=${s##*:}
    python tools/ntt_ab.py --no-check --sizes $ln --batch-sizes --variants "d$d=ACX_NTT_DIGITS=$d" 2>&1 | grep bn254
  done
done
