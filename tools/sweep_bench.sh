set -o pipefail
C="--steps 10 --warmup 2 --cpu-scans 0 --no-extras"
for args in "--k 8" "--k 16" "--k 7" "--dtype float64" "--active-only" "--no-sort" "--point-format float" "--scans 3 --points 50000" "--no-block-tables" "--bwd-layout slots" "--fwd-generic" "--autograd" "--autograd --graph"; do
  out=$(timeout -k 10 200 python3 bench.py $C $args 2>gpurun_out/sweep_err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
print('%.1f us/step  loss %.6g  fwd=%s' % (d['ms_per_step']*1e3, d['config']['final_loss'], (r.get('kernels') or {}).get('consistency_fwd',{}).get('kernel')))") || { echo "FAILED: $args"; tail -5 gpurun_out/sweep_err.log; exit 1; }
  echo "$args => $out"
done
