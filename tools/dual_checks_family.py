import pickle, math, random
from math import comb
f=pickle.load(open('gpurun_out/dual_checks.pkl','rb'))
words=[]
for w in sorted(f):
    for J,m in sorted(f[w]):
        hi=frozenset(57+k for k in range(7) if m>>k&1)
        words.append((frozenset(J), hi))
T=2
def evaluate(fam, surv_cost=120):
    k=len(fam)
    vs=[]
    for x in range(58):
        s=0
        for i,(J,hi) in enumerate(fam):
            if x in J or x in hi: s|=1<<i
        vs.append(s)
    t=max(bin(v).count('1') for v in vs)
    for a in range(58):
        for b in range(a+1,58):
            c=bin(vs[a]^vs[b]).count('1')
            if c>t: t=c
    pas=sum(comb(k,i) for i in range(min(t,k)+1))/2**k
    planes=set()
    for J,hi in fam: planes|=J
    P=len(planes)
    logic=0
    for J,hi in fam:
        logic+=math.ceil((len(J)-1)/2)*2.5 + (2.5 if len(hi)%2 else 0) + 5
    logic+=12*2.5
    return 4.3*P+logic+4*pas*surv_cost, pas, t, P, k
# max disjoint family (on positions 0..57) by randomized greedy
random.seed(7)
bestd=[]
for trial in range(20000):
    order=random.sample(range(len(words)),len(words))
    used=set(); fam=[]
    for i in order:
        J,hi=words[i]
        pos=set(J)|({57} if 57 in hi else set())
        if pos&used: continue
        used|=pos; fam.append(words[i])
    if len(fam)>len(bestd): bestd=fam
print('max disjoint family size',len(bestd), evaluate(bestd))
for J,hi in bestd: print('  ',sorted(J),sorted(hi))
# annealing over general families
def anneal(start,iters=6000,T0=30.0):
    fam=list(start); cur=evaluate(fam); best=(cur,list(fam))
    for it in range(iters):
        temp=T0*(1-it/iters)+0.01
        g=list(fam)
        r=random.random()
        if r<0.4 or len(g)<3: g.append(random.choice(words))
        elif r<0.7: g.pop(random.randrange(len(g)))
        else: g[random.randrange(len(g))]=random.choice(words)
        if len(set(g))!=len(g) or len(g)>30: continue
        e=evaluate(g)
        if e[0]<cur[0] or random.random()<math.exp((cur[0]-e[0])/temp):
            fam=g; cur=e
            if cur[0]<best[0][0]: best=(cur,list(fam))
    return best
b=anneal(bestd)
print('annealed',b[0])
for J,hi in b[1]: print('  ',sorted(J),sorted(hi))
pickle.dump(b,open('gpurun_out/dual_checks_best.pkl','wb'))
