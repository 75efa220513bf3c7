# A/B of bench.py's headline on one box: round 3's defaults against round 2's behaviour (TKSPMV_SMALL_PACKETS=0)
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys;l=json.load(open(sys.argv[1]));r=l['roofline'];print(sys.argv[2], round(l['value']), 'q/s kernel_us', round(r['kernel_us'],2), 'frac', round(r['frac'],3), 'read_only', round(r['read_only']['us_per_pass'],2), 'median', round(l.get('timing',{}).get('kernel_us_median',0),2), 'p95', round(l.get('timing',{}).get('kernel_us_p95',0),2), l['parity_checked'])" $1 "$2"; }
for i in 1 2; do
  for sp in 0 default; do
    if [ $sp = 0 ]; then export TKSPMV_SMALL_PACKETS=0; else unset TKSPMV_SMALL_PACKETS; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --skip-warm --cpu-seconds 0 --traffic off > gpurun_out/ab20.json 2>/dev/null; show gpurun_out/ab20.json "steps20 small_packets=$sp"
    python bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --skip-warm --traffic off > gpurun_out/ab3000.json 2>/dev/null; show gpurun_out/ab3000.json "steps3000 small_packets=$sp"
  done
done
