import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)
sel=rows[n//2:n//2+24]
t0=int(sel[0]['Start_Timestamp'])
for r in sel:
    print(r['Kernel_Name'][:14], 'q',r.get('Queue_Id'), 'start %.1f'%((int(r['Start_Timestamp'])-t0)/1e3), 'dur %.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
d=collections.defaultdict(list)
for r in rows: d[r['Kernel_Name'][:14]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print(k,len(v),'avg %.2f'%(sum(v)/len(v)))
