set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python3 tools/profile_paths.py --what online --reps 220 > gpurun_out/online_lat.txt 2>&1
cat gpurun_out/online_lat.txt | tail -2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/onl -o p -- python3 tools/profile_paths.py --what online --reps 220 > /tmp/onl.log 2>&1
cp $(find /tmp/onl -name 'p_kernel_stats.csv' | head -1) gpurun_out/online_kernel_stats.csv
python3 - <<EOF
import csv
tot=0; nl=0
for r in csv.DictReader(open('gpurun_out/online_kernel_stats.csv')):
    c=int(r['Calls'])/220; tot+=float(r['TotalDurationNs'])/220e3; nl+=c
    print('%5.1f x %6.1f us  %s' % (c, float(r['AverageNs'])/1e3, r['Name'][:110]))
print('total us/scan', tot, 'launches', nl)
EOF
