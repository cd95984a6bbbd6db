import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    if 'k_' in r['Name']: print("   %-26s avg %8.1f us" % (r['Name'][:26], float(r['AverageNs'])/1e3))
