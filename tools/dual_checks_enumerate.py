import numpy as np, itertools, sys, collections, pickle, time
G = 0o260534236651
col=[]; c=1
for j in range(64):
    col.append(c); c<<=1
    if c>>34 &1: c^=G
rows=[(1<<(34+b))|col[34+b] for b in range(30)]
gcol=[sum(((rows[b]>>j)&1)<<b for b in range(30)) for j in range(64)]
tg={}
for m in range(128):
    v=0
    for k in range(7):
        if m>>k&1: v^=gcol[57+k]
    tg.setdefault(v,[]).append(m)
N=57
def subsets_xor(k):
    idx=np.array(list(itertools.combinations(range(N),k)),dtype=np.int8)
    g=np.array(gcol[:N],dtype=np.int64)
    x=np.zeros(len(idx),dtype=np.int64)
    for i in range(k): x^=g[idx[:,i]]
    return idx,x
S={k:subsets_xor(k) for k in (3,4)}
found=collections.defaultdict(set)
tvals=np.array(list(tg.keys()),dtype=np.int64)
def search(ka,kb):
    ia,xa=S[ka]; ib,xb=S[kb]
    order=np.argsort(xb,kind='stable'); xbs=xb[order]
    for t in tvals:
        q=xa^t
        lo=np.searchsorted(xbs,q,'left'); hi=np.searchsorted(xbs,q,'right')
        hit=np.nonzero(hi>lo)[0]
        for h in hit:
            A=set(ia[h].tolist())
            for o in order[lo[h]:hi[h]]:
                B=set(ib[o].tolist())
                if A&B: continue
                J=tuple(sorted(A|B))
                for m in tg[int(t)]:
                    found[len(J)].add((J,m))
for ka,kb in ((3,3),(3,4),(4,4)):
    t0=time.time(); search(ka,kb); print(ka,kb,{k:len(v) for k,v in found.items()},round(time.time()-t0,1),flush=True)
pickle.dump(dict(found),open('gpurun_out/dual_checks.pkl','wb'))
