import numpy as np, sys
sys.path.insert(0,'/root/repo')
from thunder_amd.refine import pixel_list
rng=np.random.default_rng(1)
N,pf=256,2; P=N*pf
pl=pixel_list(N,N//2-2,0,pf)
ic=pl['iCol']*pf; ir=pl['iRow']*pf
nP=len(ic)
def rotmat(q):
    q0,q1,q2,q3=q
    A=np.array([[0,-q3,q2],[q3,0,-q1],[-q2,q1,0]])
    return np.eye(3)+2*q0*A+2*A@A
def randq():
    q=rng.normal(size=4); return q/np.linalg.norm(q)
def qmul(a,b):
    return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2], a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])
BLX,BLY,BLZ=4,3,3
def records(nImg=6,G=40,spread=0.01):
    """returns list of (brickkey, cx,cy,cz, img, region, pass, gl, pix)"""
    out=[]
    for img in range(nImg):
        q0=randq()
        Rs=[]
        for g in range(G):
            d=rng.normal(0,spread,3); dq=np.array([1,d[0]/2,d[1]/2,d[2]/2]); dq/=np.linalg.norm(dq)
            Rs.append(rotmat(qmul(q0,dq)))
        for g in range(G):
            R=Rs[g]
            x=R[0,0]*ic+R[0,1]*ir; y=R[1,0]*ic+R[1,1]*ir; z=R[2,0]*ic+R[2,1]*ir
            neg=x<0
            x=np.where(neg,-x,x); y=np.where(neg,-y,y); z=np.where(neg,-z,z)
            X0=np.floor(x).astype(int); yb=np.floor(y).astype(int)+P//2; zb=np.floor(z).astype(int)+P//2
            key=((zb>>BLZ)*(P>>BLY)+(yb>>BLY))*((P//2+1+15)>>BLX)+(X0>>BLX)
            pix=np.arange(nP)
            out.append(np.stack([key,X0&15,yb&7,zb&7,np.full(nP,img),pix//256,np.full(nP,g//8),np.full(nP,g%8),pix],1))
    return np.concatenate(out)
rec=records()
# stream order: sort by brick key, then (img, region, pass) [segment], then within segment (gl, pix)
order=np.lexsort((rec[:,8],rec[:,7],rec[:,6],rec[:,5],rec[:,4],rec[:,0]))
rec=rec[order]
print("records",len(rec),"bricks",len(np.unique(rec[:,0])))
def cost(idx_fn, lanegroup=32, within='gl_pix', nb=4000):
    # per brick run, batches of 64 consecutive records
    keys=rec[:,0]
    starts=np.flatnonzero(np.r_[True,keys[1:]!=keys[:-1]]); ends=np.r_[starts[1:],len(rec)]
    tot=0; ideal=0; nbat=0
    for s,e in zip(starts,ends):
        r=rec[s:e]
        if within=='cell':  # sort each segment by cell
            seg=np.lexsort((r[:,1]+16*r[:,2]+128*r[:,3], r[:,6],r[:,5],r[:,4]))
            r=r[seg]
        elif within=='pix_gl':
            seg=np.lexsort((r[:,7],r[:,8], r[:,6],r[:,5],r[:,4])); r=r[seg]
        for b in range(0,len(r),64):
            bt=r[b:b+64]
            for v in range(8):
                ii,jj,kk=v&1,(v>>1)&1,v>>2
                idx=idx_fn(bt[:,1]+ii,bt[:,2]+jj,bt[:,3]+kk)
                for g0 in range(0,len(bt),lanegroup):
                    bank=idx[g0:g0+lanegroup]%32
                    tot+=np.bincount(bank,minlength=32).max()
                    ideal+=1
            nbat+=1
            if nbat>=nb: break
        if nbat>=nb: break
    return tot/ideal
lin=lambda x,y,z:(z*9+y)*17+x
print("random baseline (32-lane groups):", np.mean([np.bincount(rng.integers(0,1377,32)%32,minlength=32).max() for _ in range(20000)]))
print("random baseline (16-lane groups):", np.mean([np.bincount(rng.integers(0,1377,16)%32,minlength=32).max() for _ in range(20000)]))
for lg in (32,16):
    print("lanegroup",lg)
    print("  current layout (17x9x9), order (gl,pix):", cost(lin,lg))
    print("  current layout, segment sorted by cell:", cost(lin,lg,'cell'))
    print("  current layout, order (pix,gl):", cost(lin,lg,'pix_gl'))
    for sx,sy in ((17,153),(18,162),(19,171),(17,155),(17,157),(16,144),(21,189),(33,297)):
        print("  strides y=%d z=%d:"%(sx,sy), cost(lambda x,y,z:z*sy+y*sx+x,lg))
