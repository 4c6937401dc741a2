import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
starts=[int(r['Start_Timestamp']) for r in rows]; ends=[int(r['End_Timestamp']) for r in rows]
i=len(rows)-1
while i>0 and starts[i]-ends[i-1] < 1_000_000: i-=1
last=rows[i:]
def nm(r): return r['Kernel_Name'].replace('void ','').replace('elag::k::(anonymous namespace)::','').replace('elag::(anonymous namespace)::','').split('(')[0][:44]
print(len(last),'kernels, span %.2f ms busy %.2f ms'%((ends[-1]-starts[i])/1e6,sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in last)/1e6))
t0=starts[i]
for r in last:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if d>=40: print('%8.2f ms  %7.1f us  %s'%((int(r['Start_Timestamp'])-t0)/1e6,d,nm(r)))
