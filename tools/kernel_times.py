import csv, glob, sys
f = glob.glob('/tmp/fl/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    n = r['Kernel_Name']
    if any(k in n for k in sys.argv[1:]):
        print("%-40s %8.3f ms" % (n.split('(')[0][-40:], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
