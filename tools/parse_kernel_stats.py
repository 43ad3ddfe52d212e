import csv,sys,glob
for d in sys.argv[1:]:
    f=glob.glob(d+'/*/*kernel_stats.csv')[0]
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print("==",d,"total ms %.2f"%(tot/1e6))
    for r in rows[:14]:
        print("  %-58s calls %4s avg %8.1f us %5.1f %%"%(r['Name'][:58],r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
