# 3M vs 4M cross-spectral kernels over channel counts (development aid): bash tools/csd_channels_probe.sh 128 64 32
for c in "$@"; do
  for m in 3M 4M; do
    if [ $m = 4M ]; then export SPYHIP_CSD_4M=1; else unset SPYHIP_CSD_4M; fi
    python bench.py --channels $c --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('C=$c $m', round(r['value']), 'trials/s', round(r['roofline']['achieved'],1), 'TF', round(r['roofline']['frac'],3), r['roofline']['kernel'][:34], 'csd us/trial', round(1e3*r['config']['csd_ms_per_trial'],2), 'fft us/trial', round(1e3*r['config']['fft_ms_per_trial'],2))"
  done
done
