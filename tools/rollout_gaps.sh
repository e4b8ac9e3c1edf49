cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/gap && mkdir -p $R/gpurun_out/gap
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gap -o ro -- python $R/tools/rollout_once.py 4 > /dev/null 2>&1
python - <<'PY'
import csv, os, glob
fn = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/gap/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last rollout: find last init_state_kernel
idx = [i for i, n in enumerate(names) if 'init_state' in n]
a = idx[-1]
seg = rows[a:]
import collections
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for p, c in zip(seg[:-1], seg[1:]):
    g = (int(c['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3
    key = (p['Kernel_Name'].split('(')[0][-28:], c['Kernel_Name'].split('(')[0][-28:])
    gaps[key].append(g)
for r in seg:
    durs[r['Kernel_Name'].split('(')[0][-28:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot_gap = sum(sum(v) for v in gaps.values()); tot_dur = sum(sum(v) for v in durs.values())
print('kernels', len(seg), 'sum durations %.1f us' % tot_dur, 'sum gaps %.1f us' % tot_gap, 'span %.1f us' % ((int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print('gap', k, 'n=%d mean %.2f us' % (len(v), sum(v) / len(v)))
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print('dur', k, 'n=%d mean %.2f us' % (len(v), sum(v) / len(v)))
PY
rm -rf $R/gpurun_out/gap
