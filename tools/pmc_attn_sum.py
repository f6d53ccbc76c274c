import csv,sys,collections,glob
for f in sorted(glob.glob(sys.argv[1])):
    acc=collections.defaultdict(lambda: [0,0])
    for r in csv.DictReader(open(f)):
        if 'enc_attention' in r['Kernel_Name']:
            acc[r['Counter_Name']][0]+=float(r['Counter_Value']); acc[r['Counter_Name']][1]+=1
    for k,(v,n) in acc.items(): print(f.split('/')[-1], k, '%.4g per launch (%d launches)'%(v/max(n,1),n))
