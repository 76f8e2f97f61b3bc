import numpy as np, sys
d = np.load('/tmp/sim/state0.npz')
m2, co, pl, rg, nc = d['means2D'], d['conic_opacity'], d['point_list'], d['ranges'], d['n_contrib']
gx=64
rng = np.random.default_rng(0)
tiles = [t for t in range(4096) if rg[t,1]>rg[t,0]]
sel = rng.choice(tiles, size=120, replace=False)
REGX = int(sys.argv[1]); REGY=int(sys.argv[2]); BS=int(sys.argv[3])
res = {}
tot = dict(regions=0, walked=0, surv=0, pairs_act=0, pairs_mask=0, batches=0)
for NB in (2,3,4,5,6,8,1000):
    res[NB]=0
for t in sel:
    ty, tx = divmod(t, gx)
    ids = pl[rg[t,0]:rg[t,1]]
    L = len(ids)
    x = m2[ids,0]; y = m2[ids,1]; a=co[ids,0]; b=co[ids,1]; c=co[ids,2]; op=co[ids,3]
    qcut = (2*np.log(np.maximum(255*op,1e-30))+0.01)*1.01
    pos = np.arange(1, L+1)
    for ry in range(0,16,REGY):
        for rx in range(0,16,REGX):
            py, px = np.meshgrid(np.arange(ty*16+ry, ty*16+ry+REGY), np.arange(tx*16+rx, tx*16+rx+REGX), indexing='ij')
            py=py.ravel(); px=px.ravel()
            ncp = nc[py,px]
            wmax = ncp.max()
            if wmax==0: continue
            dx = x[:wmax,None]-px[None,:]; dy = y[:wmax,None]-py[None,:]
            q = a[:wmax,None]*dx*dx + 2*b[:wmax,None]*dx*dy + c[:wmax,None]*dy*dy
            alpha = np.minimum(0.99, op[:wmax,None]*np.exp(-0.5*q))
            inlist = pos[:wmax,None] <= ncp[None,:]
            act = (q>=0)&(alpha>=1/255)&inlist
            mask0 = (q<=qcut[:wmax,None])
            surv = mask0.any(1)
            ms0 = mask0[surv][::-1]; il = inlist[surv][::-1]
            nb = (len(ms0)+BS-1)//BS
            # batch-granular inlist: batch fully invalid for pixel if its LAST slot (smallest pos) still > nc
            ms = ms0.copy()
            for bi in range(nb):
                blk = slice(bi*BS, min((bi+1)*BS, len(ms0)))
                allinvalid = ~il[blk][-1]     # last slot of batch (frontmost) not in list -> none is
                ms[blk][:, allinvalid] = False
                allvalid = il[blk][0]
                mixed = ~allinvalid & ~allvalid
                # mixed: exact (binary search)
                ms[blk] = np.where(mixed[None,:], ms0[blk] & il[blk], ms[blk])
            tot['regions']+=1; tot['walked']+=wmax; tot['surv']+=surv.sum(); tot['pairs_act']+=act.sum(); tot['pairs_mask']+=ms.sum(); tot['batches']+=nb
            cnt = np.stack([ms[bi*BS:(bi+1)*BS].sum(0) for bi in range(nb)]) if nb else np.zeros((0,len(px)),int)  # [nb, lanes]
            for NB in res:
                # simulate
                lanes = cnt.shape[1]
                cur = np.zeros(lanes, int)      # current batch per lane
                rem = cnt[0].copy() if nb else np.zeros(lanes,int)
                produced = 1 if nb else 0
                it = 0
                while True:
                    # advance lanes with rem==0 to next produced batch
                    moved=True
                    while moved:
                        adv = (rem==0)&(cur<produced-1)
                        moved = adv.any()
                        if moved:
                            cur[adv]+=1; rem[adv]=cnt[cur[adv], np.nonzero(adv)[0]]
                    oldest = cur.min() if (rem>0).any() or produced<nb else produced
                    # lanes finished with everything produced sit at cur=produced-1, rem=0 -> they don't hold old batches
                    hold = np.where(rem>0, cur, produced)   # lane holds its current batch only if it has work there
                    oldest = hold.min()
                    if produced<nb and produced-oldest<NB:
                        produced+=1; continue
                    if not (rem>0).any():
                        if produced>=nb: break
                        produced+=1; continue
                    rem[rem>0]-=1; it+=1
                res[NB]+=it
R=tot['regions']
print({k:v/R for k,v in tot.items()})
lanes=REGX*REGY
for NB,n in res.items():
    print('NB',NB,'iters/region',n/R,'util(act)',tot['pairs_act']/(lanes*n),'util(mask)',tot['pairs_mask']/(lanes*n))
