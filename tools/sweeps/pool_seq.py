import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
seq=collections.defaultdict(list)
keys=('moi_pool_fwd','moi_pool_bwd_busy','align_bwd_gather','tile_plan','fillBuffer','moi_pool_bwd_tiled','align_bwd_tiled','moi_cell_bits','moi_roi_bits','moi_tile_census','moi_roi_lists','align_census_kernel','moi_fwd_rows','moi_bits_all','moi_pool_bwd_levels','align_bwd_nhwc')
for r in rows:
    n=r['Kernel_Name']
    for key in keys:
        if key in n:
            seq[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in seq.items():
    # compress runs
    out=[]; 
    for x in v:
        x=round(x)
        if out and abs(out[-1][0]-x)<=max(3,0.05*x): out[-1][1]+=1
        else: out.append([x,1])
    print(k, len(v), ' '.join('%dx%d'%(c,x) for x,c in out[:40]))
