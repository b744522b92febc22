import sqlite3,sys,collections
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name,start,duration from kernels order by start").fetchall()
last=[i for i,r in enumerate(rows) if 'k_sh_w' in r[0]]
beg=last[-2]+1; end=last[-1]+1
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows[beg:end]:
    n=r[0].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    agg[n][0]+=1; agg[n][1]+=r[2]/1e6
tot=sum(v[1] for v in agg.values())
for n,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:22]:
    print("%-34s calls %4d  %8.3f ms  %5.1f%%"%(n[-34:],v[0],v[1],100*v[1]/tot))
print("total kernel ms",tot,"span ms",(rows[end-1][1]-rows[beg][1])/1e6)
