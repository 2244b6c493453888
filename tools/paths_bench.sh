#!/bin/bash
# The other ways into the hot path, at HEAD, one JSON line each (on the GPU box, from the repo root):
#   tools/paths_bench.sh <out_dir>
# native loop, drop-in loop (autograd Function + torch.optim.Adam), the same replayed as one hipGraph, the native loop with
# the RCCL all-reduce on one rank (its overhead), the basis form with separate forward / backward kernels, the general path
# (no basis form), and a C4-shaped train() iteration.
set -o pipefail
OUT=${1:?out dir}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
COMMON="--steps 200 --warmup 20 --cpu-scans 0 --no-extras"
python3 bench.py $COMMON > "$OUT/bench_native.json" 2> "$OUT/bench_native.err" || exit 1
python3 bench.py $COMMON --no-chain > "$OUT/bench_no_chain.json" 2> "$OUT/bench_no_chain.err" || exit 1
python3 bench.py $COMMON --two-pass > "$OUT/bench_two_pass.json" 2> "$OUT/bench_two_pass.err" || exit 1
python3 bench.py $COMMON --no-basis > "$OUT/bench_general.json" 2> "$OUT/bench_general.err" || exit 1
python3 bench.py $COMMON --autograd > "$OUT/bench_autograd.json" 2> "$OUT/bench_autograd.err" || exit 1
python3 bench.py $COMMON --autograd --dc-adam > "$OUT/bench_autograd_dcadam.json" 2> "$OUT/bench_autograd_dcadam.err" || exit 1
python3 bench.py $COMMON --autograd --graph > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.err" || exit 1
python3 bench.py $COMMON --autograd --graph --dc-adam > "$OUT/bench_graph_dcadam.json" 2> "$OUT/bench_graph_dcadam.err" || exit 1
DC_FORCE_DIST=1 python3 bench.py $COMMON > "$OUT/bench_dist1.json" 2> "$OUT/bench_dist1.err" || exit 1
DC_FORCE_DIST=1 python3 bench.py $COMMON --no-chain > "$OUT/bench_dist1_no_chain.json" 2> "$OUT/bench_dist1_no_chain.err" || exit 1
python3 tools/c4_bench.py > "$OUT/c4.json" 2> "$OUT/c4.err" || exit 1
python3 - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
for name in ('bench_native', 'bench_no_chain', 'bench_two_pass', 'bench_general', 'bench_autograd', 'bench_autograd_dcadam', 'bench_graph',
             'bench_graph_dcadam', 'bench_dist1', 'bench_dist1_no_chain'):
    d = json.loads(open(os.path.join(out, name + '.json')).read().strip().splitlines()[-1])
    print('%-16s %.1f us/step  %.3g points/s  (%s)' % (name, d['ms_per_step'] * 1e3, d['value'], d['config']['loop']))
print(open(os.path.join(out, 'c4.json')).read().strip().splitlines()[-1][:400])
PY
