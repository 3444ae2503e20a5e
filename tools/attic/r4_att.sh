# average duration of the tower's attention kernel under rocprofv3 (ViT-L/14, batch 32): bash tools/r4_att.sh
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/a_att -o k -- python $R/tools/perf_clip.py 32 L/14 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/a_att/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'attention' in r['Name']: print('%s avg %.1f us (calls %s)' % (r['Name'][:40], float(r['AverageNs'])/1e3, r['Calls']))
PY
