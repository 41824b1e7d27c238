import csv,glob,sys
rows=[]
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dec_' in r['Kernel_Name']: rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][:44],r.get('Grid_Size_X',r.get('Grid_Size',''))))
rows.sort()
last=rows[-26:]
for s,e,n,g in last: print("%-46s %9s %7.1f us" % (n,g,(e-s)/1e3))
