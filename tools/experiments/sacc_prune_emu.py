import numpy as np, sys
rng=np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 1)
NS=16; GAS=(NS-6)%NS
def run(k,nt,G,chunk,e,out_lo,out_hi):
    st=[int(np.floor(g*k))-nt//2 for g in range(G)]
    for g in range(G-1): assert st[g+1]>st[g]
    # host check
    for g in range(G-NS):
        gend=((st[g]+nt-1)//4)*4+3
        if not st[g+NS]>gend: return None
    u_min=st[0]; u_max=st[-1]+nt-1
    tbase=(u_min//4)*4
    coef=rng.rand(G,nt).astype(np.float32)+0.5
    smp={u:np.float32(rng.rand()+0.5) for u in range(tbase-8,u_max+16)}
    def row(u):
        r=np.zeros(NS,np.float32)
        for g in range(G):
            if st[g]<=u<=st[g]+nt-1: r[g%NS]=coef[g][u-st[g]]
        return r
    def lastq(q):
        la=-1
        while la<G-1 and st[la+1]<=4*q+3: la+=1
        return la
    def first_alive(u):
        for g in range(G):
            if st[g]+nt-1>=u: return g
        return G-1
    ref=np.zeros(G,np.float32)
    for g in range(G):
        a=np.float32(0)
        for t in range(nt): a=np.float32(a+np.float32(coef[g][t]*smp[st[g]+t]))
        ref[g]=a
    md=((out_lo+chunk+3-e)%NS+NS)%NS
    cbase=out_lo-md
    nchunks=(out_hi-cbase+chunk-1)//chunk
    bad=0; npro=0; ntail=0
    for c in range(nchunks):
        n0c=cbase+c*chunk; n0=max(n0c,out_lo); n1=min(n0c+chunk,out_hi)
        if n0>=n1: continue
        clamp=lambda v:max(0,min(v,G-1))
        ga=clamp(n0-e-3); gb=clamp(n1-1-e+3); gbv=n1-1-e+3
        u_a=st[ga]; q_a=u_a>>2
        g_al=first_alive(u_a)
        pro=(ga%NS==GAS and g_al<ga)
        tail_g=gb-(NS-1) if gb%NS==NS-1 else 10**9
        R=np.zeros(NS,np.float32)
        Tg=ga if pro else g_al
        endg=st[Tg]+nt-1
        qn=q_a; ulast=4*q_a-1
        def block(mask,skiplow=0):
            nonlocal qn,ulast
            for j in range(4):
                u=4*qn+j
                x=smp[u] if u>=u_a else np.float32(0)
                r=row(u)
                for q in range(4):
                    if (mask>>q)&1 and q>=skiplow:
                        for s in range(4*q,4*q+4):
                            R[s]=np.float32(R[s]+np.float32(x*r[s]))
            qn+=1; ulast+=4
        if pro:
            npro+=1
            bf=(st[ga-1]+nt-1)>>2
            while qn<=bf:
                m=lastq(qn)-ga+1
                rk=(GAS%4+m-1)>>2
                NQ=min(max(rk,0),3)+1
                mask=0
                for r_ in range(NQ): mask|=1<<((GAS//4+r_)%4)
                block(mask)
                for i in range(1,NS):
                    r_=(GAS%4+i)>>2
                    if (r_==NQ-1 or r_>=4) and i>=m:
                        R[(GAS+i)%NS]=0
        out={}
        while True:
            S=Tg%NS
            while Tg<=gb and endg>ulast:
                if S//4>0 and Tg>=tail_g: ntail+=1; block(15,S//4)
                else: block(15)
            val=R[S]; R[S]=0
            if Tg>=ga and Tg<=gb: out[Tg]=val
            Tg+=1
            if Tg<=gb: endg=st[Tg]+nt-1
            if Tg>gbv: break
        for g in range(ga,gb+1):
            if out[g]!=ref[g]: bad+=1
    return bad,npro,ntail
tot=0
for it in range(40):
    k=rng.uniform(2.0,6.0)
    if rng.rand()<0.4: k=float(rng.choice([2.5,2.7,3.0,3.5,4.0]))
    nt=int(12.2*k)+rng.randint(0,3)
    G=rng.randint(40,200); chunk=16*rng.randint(1,4); e=rng.randint(0,4)
    lo=rng.randint(0,10); hi=G-rng.randint(0,10)
    r=run(k,nt,G,chunk,e,lo,hi)
    print(it,round(k,3),nt,G,chunk,e,r)
