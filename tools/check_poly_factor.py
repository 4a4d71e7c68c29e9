#!/usr/bin/env python3
"""Round 6: could the sliding check stream be computed in two stages?  Every multiple q of the reversed cofactor with deg <= 37 is a valid
check polynomial; if q = a * b the stream is (stream * a) * b: (wt(a) - 1) funnel shifts for two or three dwords of the intermediate
plus (wt(b) - 1) for two dwords, instead of wt(q) - 1 for two.  Prices every factorisation of every candidate with the issue cycles of
tools/valu_rate.hip (funnel shift 4.2, v_bitop3 2.6).  Result: best 96.4 cycles per word (q of weight 13 = (1 + x + x^8) * 7 taps)
against 110 for the lightest q (11 taps, what slide.h picks): 2 % of the launch, before the planes shared with the barker filter are
counted against it.  Not built.  CPU only: python tools/check_poly_factor.py"""
import itertools
def deg(p): return p.bit_length()-1
def clmul(a,b):
    r=0
    while b:
        if b&1: r^=a
        a<<=1; b>>=1
    return r
def pdivmod(a,b):
    q=0; db=deg(b)
    while a and deg(a)>=db:
        s=deg(a)-db; q|=1<<s; a^=b<<s
    return q,a
def rev(p):
    d=deg(p); return sum(((p>>i)&1)<<(d-i) for i in range(d+1))
g=0o260534236651
h,rem=pdivmod((1<<63)|1,g); assert rem==0
hr=rev(h)
print("deg g",deg(g),"deg h",deg(h),"wt hr",bin(hr).count('1'))
# irreducible factorization by trial division
def factor(p):
    fs=[]; d=2
    while deg(p)>0:
        if deg(p)<2*deg(d) if d>1 else False:
            fs.append(p); break
        q,r=pdivmod(p,d)
        if r==0: fs.append(d); p=q
        else: d+=1
    return fs
def taps(p): return [k for k in range(deg(p)+1) if (p>>k)&1]
def nshift(p, allow32=True):
    return sum(1 for k in taps(p) if k%32!=0)
def nxor(w):  # xor3 count for w planes
    return (w-1+1)//2
best=[]
span=37
for a in range(1,1<<(span-deg(hr)+1),2):
    q=clmul(hr,a)
    if deg(q)>span: continue
    fs=factor(q)
    n=len(fs)
    wq=bin(q).count('1')
    base_cost=2*( (wq-1)*4.2 + nxor(wq)*2.6 )
    seen=set()
    for r in range(1,n):
        for idx in itertools.combinations(range(n),r):
            A=1
            for i in idx: A=clmul(A,fs[i])
            B,rm=pdivmod(q,A); assert rm==0
            if (A,B) in seen: continue
            seen.add((A,B))
            for sh_on in (0,1):   # where the overall <<1 goes
                a1=A<<(1 if sh_on==0 else 0); b1=B<<(1 if sh_on==1 else 0)
                wa=bin(A).count('1'); wb=bin(B).count('1')
                for nu in (2,3):   # dwords of the intermediate computed per lane (2: third via bpermute)
                    cost=nu*(nshift(a1)*4.2+nxor(wa)*2.6)+2*(nshift(b1)*4.2+nxor(wb)*2.6)
                    best.append((cost,nu,hex(q),wq,taps(a1),taps(b1),base_cost))
best.sort()
for b in best[:25]: print(b)
# baseline: current q
cur=min((bin(clmul(hr,a)).count('1'),clmul(hr,a)) for a in range(1,1<<9,2) if deg(clmul(hr,a))<=span)
print("lightest",cur[0],taps(cur[1]<<1))
