import sqlite3, sys, collections
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
frames=int(sys.argv[2]) if len(sys.argv)>2 else 400
rows=list(cur.execute("select name, start, end from kernels order by start"))
res=[i for i,r in enumerate(rows) if 'k_resolve' in r[0]]
first=res[-frames-1]+1
sel=rows[first:]
t0=sel[0][1]; t1=sel[-1][2]
agg=collections.defaultdict(lambda:[0,0])
for n,s,e in sel:
    k=n.split('(')[0][:60]; agg[k][0]+=1; agg[k][1]+=e-s
tot=0
for k,(c,t) in sorted(agg.items(), key=lambda x:-x[1][1]):
    print(f"{k:62s} {c:6d} {t/c/1e3:9.2f} us avg  {t/1e6:8.2f} ms total"); tot+=t
iv=sorted((s,e) for n,s,e in sel); busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print(f'window {(t1-t0)/1e6:.2f} ms, per frame {(t1-t0)/frames/1e3:.1f} us, kernel sum per frame {tot/frames/1e3:.1f} us, busy {busy/(t1-t0):.3f}')
