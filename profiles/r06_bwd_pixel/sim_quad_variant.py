import numpy as np
d = np.load('/tmp/sim/state0.npz')
m2, co, pl, rg, nc = d['means2D'], d['conic_opacity'], d['point_list'], d['ranges'], d['n_contrib']
gx=64
rng = np.random.default_rng(1)
tiles = [t for t in range(4096) if rg[t,1]>rg[t,0]]
sel = rng.choice(tiles, size=80, replace=False)
def rect_min_q(x,y,a,b,c, X0,X1,Y0,Y1):
    # min of q over rectangle of pixel centres by brute force over the pixel centres (integer grid) -- close to the exact continuous test
    best = np.full(x.shape, np.inf)
    for py in range(Y0, Y1+1):
        for px in range(X0, X1+1):
            dx = x-px; dy = y-py
            best = np.minimum(best, a*dx*dx+2*b*dx*dy+c*dy*dy)
    return best
tot = dict(blocks=0, steps_now=0, steps_drain=0, steps_carry=0, steps_ideal=0, surv=0, qsurv=0, act=0)
for t in sel:
    ty, tx = divmod(t, gx)
    ids = pl[rg[t,0]:rg[t,1]]
    x = m2[ids,0]; y = m2[ids,1]; a=co[ids,0]; b=co[ids,1]; c=co[ids,2]; op=co[ids,3]
    qcut = (2*np.log(np.maximum(255*op,1e-30))+0.01)*1.01
    L=len(ids)
    for by in range(4):
        for bx in range(4):
            X0=tx*16+bx*4; Y0=ty*16+by*4
            ncb = nc[Y0:Y0+4, X0:X0+4]
            wmax = int(ncb.max())
            if wmax==0: continue
            sl = slice(0,wmax)
            keep_blk = rect_min_q(x[sl],y[sl],a[sl],b[sl],c[sl],X0,X0+3,Y0,Y0+3) <= qcut[sl]
            kq = []
            for qy in range(2):
                for qx in range(2):
                    kq.append(rect_min_q(x[sl],y[sl],a[sl],b[sl],c[sl],X0+qx*2,X0+qx*2+1,Y0+qy*2,Y0+qy*2+1) <= qcut[sl])
            kq = np.array(kq)   # [4, wmax]
            # walk from the back in passes of 64
            order = np.arange(wmax-1,-1,-1)
            tot['blocks']+=1
            nb = keep_blk.sum(); tot['surv']+=nb; tot['qsurv']+=kq.sum()
            tot['steps_now'] += (nb+3)//4
            tot['steps_ideal'] += max((kq[q].sum()+3)//4 for q in range(4))
            pend = np.zeros(4,int); sd=0; sc=0
            npass = (wmax+63)//64
            for pi in range(npass):
                seg = order[pi*64:(pi+1)*64]
                add = kq[:,seg].sum(1)
                # drain per pass
                sd += max((add+3)//4)
                # carry: process full steps only, leftovers stay (last pass drains)
                pend += add
                if pi==npass-1:
                    sc += max((pend+3)//4); pend[:]=0
                else:
                    k = max(pend//4); sc += k; pend -= np.minimum(pend//4, k)*4
            tot['steps_drain']+=sd; tot['steps_carry']+=sc
B=tot['blocks']
print({k:v/B for k,v in tot.items()})
