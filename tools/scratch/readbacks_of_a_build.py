import csv,glob,collections,sys
import numpy as np
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
starts=[int(r['Start_Timestamp']) for r in rows]; ends=[int(r['End_Timestamp']) for r in rows]
i=len(rows)-1
while i>0 and starts[i]-ends[i-1] < 400_000: i-=1
last=rows[i:]
def nm(r): return r['Kernel_Name'].replace('void ','').replace('elag::k::(anonymous namespace)::','').replace('elag::(anonymous namespace)::','').split('(')[0][:40]
print(len(last),'kernels, span %.2f ms busy %.2f ms'%((ends[-1]-starts[i])/1e6,sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in last)/1e6))
c=collections.Counter(); seq=[]
for j,r in enumerate(last):
    if 'fetch_publish' in r['Kernel_Name']:
        prev=[nm(x) for x in last[max(0,j-2):j]]
        c[' <- '.join(reversed(prev))]+=1
        if j+1<len(last): seq.append((int(last[j+1]['Start_Timestamp'])-int(r['End_Timestamp']))/1e3)
for k,v in c.most_common(30): print(v,k)
print('read-backs',len(seq)+1,'gap after (us): median %.1f mean %.1f total %.0f'%(np.median(seq),np.mean(seq),np.sum(seq)))
agg=collections.Counter()
for r in last: agg[nm(r)]+=1
print(agg.most_common(14))
