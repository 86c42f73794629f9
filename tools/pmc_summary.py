import csv, collections, glob, sys
for d in sys.argv[1:]:
    f = (glob.glob(d+'/*counter_collection.csv') + glob.glob(d+'/*/*counter_collection.csv'))[0]
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
    for r in rows:
        k = r['Kernel_Name']; k = k[:k.index('(')] if '(' in k else k
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k,v in agg.items():
        if 'bdpt' not in k: continue
        s = f"{k[11:50]:40s} n={len(disp[k]):3d} " + " ".join(f"{n}={x:.3g}" for n,x in v.items())
        if 'SQ_ACTIVE_INST_VALU' in v:
            s += f" | lane_util={v['SQ_THREAD_CYCLES_VALU']/(v['SQ_ACTIVE_INST_VALU']*64):.3f} valu_active/wavecyc={v['SQ_ACTIVE_INST_VALU']/v['SQ_WAVE_CYCLES']:.3f} wait_any={v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.3f} wait_inst={v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:.3f} busy_cyc/wave={v['SQ_BUSY_CYCLES']/v['SQ_WAVES']:.0f}"
        if 'TCC_HIT_sum' in v:
            s += f" | L2hit={v['TCC_HIT_sum']/(v['TCC_HIT_sum']+v['TCC_MISS_sum']):.3f} L1miss~={v['TCP_TCC_READ_REQ_sum']/max(1,v['TCP_TOTAL_CACHE_ACCESSES_sum']):.3f}"
        print(s)
