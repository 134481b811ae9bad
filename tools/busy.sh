#!/bin/sh
# GPU busy share (union of all kernel intervals) of a bench configuration under the kernel trace: sh tools/busy.sh <config> [flags]
cfg=${1:-2}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/p_busy
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_busy -- python3 $R/bench.py --config $cfg --no-cpu-baseline --no-side-legs --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/p_busy/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][:40]) for r in rows)
# last 60 % of the trace = timed window roughly
t0=iv[0][0]; t1=max(b for a,b,_ in iv); cut=t0+(t1-t0)*0.5
iv=[x for x in iv if x[0]>=cut]
busy=0; end=-1
for a,b,_ in iv:
    if b>end: busy+=b-max(a,end); end=b
span=max(b for a,b,_ in iv)-iv[0][0]
print('kernels',len(iv),'span ms',span/1e6,'busy share',round(busy/span,3))
c=collections.Counter(); n=collections.Counter()
for a,b,k in iv: c[k]+=b-a; n[k]+=1
for k,v in c.most_common(12): print(f'  {k:42s} {v/span*100:5.1f}%  n={n[k]}  avg {v/n[k]/1e3:.1f} us')
PY
