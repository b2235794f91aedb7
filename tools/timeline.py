"""Per-queue run-length summary of one training step from tools/profile_bench.sh's last_step_dispatches.csv.

    python tools/timeline.py gpurun_out/<tag>/last_step_dispatches.csv [min_segment_us]
"""
import csv,re,sys
f=sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/tl/last_step_dispatches.csv'
rows=list(csv.DictReader(open(f)))
for r in rows: r['s']=float(r['start_us']); r['d']=float(r['duration_us'])
ad=[i for i,r in enumerate(rows) if 'adam' in r['kernel']]
def short(k):
    k=k.replace('void ','')
    m=re.match(r'(\w+)(<[^>]*>)?',k); return (m.group(1)+(m.group(2) or ''))[:34]
# step k-1 = between adam[-5] and adam[-3]
a0,a1=(ad[-5],ad[-3]) if len(ad)>=5 else (ad[-4],ad[-2])
seg=rows[a0+1:a1+1]
t0=seg[0]['s']
print("step span %.1f us"%(seg[-1]['s']+seg[-1]['d']-t0))
minlen=float(sys.argv[2]) if len(sys.argv)>2 else 0
for q in sorted(set(r['queue'] for r in seg)):
    print('=== queue',q)
    cur=None
    def flush(cur):
        if cur and (cur[2]-cur[3])>=minlen: print('  %-36s x%-4d %8.1f -> %8.1f  busy %7.1f'%(cur[0],cur[1],cur[3]-t0,cur[2]-t0,cur[4]))
    for r in seg:
        if r['queue']!=q: continue
        k=short(r['kernel'])
        if cur and cur[0]==k and r['s']-cur[2]<60:
            cur[1]+=1; cur[2]=r['s']+r['d']; cur[4]+=r['d']
        else:
            flush(cur)
            cur=[k,1,r['s']+r['d'],r['s'],r['d']]
    flush(cur)
