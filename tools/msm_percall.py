import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name,start,duration,grid_x from kernels order by start").fetchall()
last=[i for i,r in enumerate(rows) if 'k_sh_w' in r[0]]
beg=last[-2]+1; end=last[-1]+1
cur=None
for r in rows[beg:end]:
    n=r[0].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')
    if n.startswith('k_msm_'):
        if n=='k_msm_hist': cur={}; 
        if cur is None: cur={}
        cur[n[6:]]=r[2]/1e3
        if n=='k_msm_weighted':
            print(' '.join('%s=%.0f'%(k,v) for k,v in cur.items()), ' total=%.0f'%sum(cur.values())); cur=None
print("sum kernel ms", sum(r[2] for r in rows[beg:end])/1e6)
