import csv, sys, glob
d=sys.argv[1]
ev=[]
for f in glob.glob(d+'/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K '+r['Kernel_Name'].replace('mtg::','').replace('(anonymous namespace)::','').replace('void ','')[:60]))
for f in glob.glob(d+'/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C '+r.get('Direction','?')+' '+r.get('Bytes', r.get('Size','?'))))
ev.sort()
idx=[i for i,e in enumerate(ev) if 'classify_kernel' in e[2]]
i=idx[-1]; t0=ev[i][0]
for e in ev[max(0,i-14):i+6]:
    print(f"{(e[0]-t0)/1e3:12.1f} {(e[1]-e[0])/1e3:10.1f}  {e[2]}")
