import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0],r['Queue_Id']) for r in rows]
rows.sort()
idx=[i for i,r in enumerate(rows) if r[2]=='k_resolve']
st=idx[-1]
t0=rows[st][0]
for r in rows[st:]:
    print("%8.3f %8.3f %7.3f q%s %s"%((r[0]-t0)/1e6,(r[1]-t0)/1e6,(r[1]-r[0])/1e6,r[3],r[2]))
