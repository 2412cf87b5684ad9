"""Summarise an MI_DEBUG_OPTIME=1 log (per-op wall times of the MatterGen-shaped program, stream-synchronised) by layer class.
usage: optime_summary.py LOG [EVALS_IN_LOG]  -- the last evaluation of the log is summarised."""
import sys,re,collections
lines=[l for l in open(sys.argv[1]) if l.startswith('[optime]')]
n=len(lines)//int(sys.argv[2]) if len(sys.argv)>2 else len(lines)
tot=collections.defaultdict(float); cnt=collections.defaultdict(int)
for l in lines[-n:]:
    m=re.match(r'\[optime\] (.*?)\s+\[(\d+) x (\d+) x (\d+)\]\s+([\d.]+) us',l)
    if not m: continue
    name,M,N,K,us=m.group(1),int(m.group(2)),int(m.group(3)),int(m.group(4)),float(m.group(5))
    if name.startswith('dense'):
        key='dense E-level N=%d K=%d'%(N,K) if M>100000 else 'dense node-level'
        if M>100000 and N==512 and K==512:
            key+=' '+('x.1 (residual)' if re.search(r'\.\d+\.1\.weight',name) else 'concat' if 'concat' in name else 'dense_ca' if 'dense_ca' in name else 'dense_ba' if 'dense_ba' in name else 'plain')
    else: key=name.split('[')[0].strip().split(' out_')[0]
    tot[key]+=us; cnt[key]+=1
T=sum(tot.values())
for k,v in sorted(tot.items(),key=lambda x:-x[1]): print('%-50s %3d calls %8.1f us  avg %7.1f  %5.1f%%'%(k,cnt[k],v,v/cnt[k],100*v/T))
print('total',T)
