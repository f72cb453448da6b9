import numpy as np
rng=np.random.default_rng(0)
N=2000;k=5;R=20000
logits=rng.normal(size=N); logits[N//2:]+=3.0
w=np.exp(logits.max()-logits)           # key = w*log2(u) ; |key| = w * E/ln2 , E~Exp(1)
cp=np.log(2)*np.sum(1/w)
def run(lam0, maxp, rule):
    probes=[];nc=[];fb=0;cl=[]
    for r in range(R):
        a=w*rng.exponential(size=N)/np.log(2)     # |key|
        pad=np.full(2048,np.inf); pad[:N]=a
        # element n=4*(lane+64*i)+j -> lane = (n//4)%64
        lane=(np.arange(2048)//4)%64
        lm=np.full(64,np.inf); np.minimum.at(lm,lane,pad)
        lam=lam0; lo=None; hi=None; clo=None; p=0; ok=False
        while p<maxp:
            t=lam/cp; c=int((lm<=t).sum()); p+=1
            if c==k: lo=lam; clo=c; ok=True; break
            if c>k:
                if lo is None or lam<lo: lo=lam; clo=c
            else:
                if hi is None or lam>hi: hi=lam
            if lo is not None and hi is not None: lam=0.5*(lo+hi)
            else: lam=lam*rule(c)
        probes.append(p)
        if lo is None or clo>12: fb+=1; continue
        t=lo/cp; nc.append(int((a<=t).sum())); cl.append(clo)
    nc=np.array(nc)
    return np.mean(probes), fb/R, np.mean(nc), np.mean(nc==k), np.percentile(nc,99)
for lam0 in (5.0,5.5,6.0,6.5):
  for maxp in (3,4,6):
    print(lam0,maxp, run(lam0,maxp,lambda c: min(4.0,(k+0.5)/(c+0.5))))
