import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=float(sys.argv[2])
tot=0
for r in rows:
    if 'kbn' in r['Name']:
        t=float(r['TotalDurationNs'])/n/1e6; tot+=t
        if 'k_reduce' in r['Name'] or 'finalize' in r['Name']:
            print('  %-46s %5.1f/step avg %6.1f us  %.3f ms/step'%(r['Name'].split('(')[0].replace('void rigl::kbn::','')[:46], int(r['Calls'])/n, float(r['AverageNs'])/1e3, t))
print('BN total ms/step %.3f'%tot)
