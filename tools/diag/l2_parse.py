import csv, collections, sys
for d in sys.argv[1:]:
    rows=list(csv.DictReader(open(f'{d}/out_counter_collection.csv')))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'p8_kernel<0' not in k: continue
        n=len(v['TCC_HIT_sum']); h=sum(v['TCC_HIT_sum'])/n; m=sum(v['TCC_MISS_sum'])/n
        print(f"{d.split('/')[-1]:28s} hit {h/1e6:8.2f}M miss {m/1e6:8.2f}M rate {h/(h+m):.3f}")
