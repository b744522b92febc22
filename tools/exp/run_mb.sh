cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_w; rocprofv3 --kernel-trace -d /tmp/prof_w -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-seconds 0 2>/dev/null | tail -1 | cut -c1-100
python - <<'PY'
import sqlite3
db=sqlite3.connect('/tmp/prof_w/r_results.db')
cols=[r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
rows=db.execute("select name,start,end,stream_id,queue_id,grid_x from kernels order by start").fetchall() if 'stream_id' in cols else db.execute("select name,start,end,0,queue_id,grid_x from kernels order by start").fetchall()
thi=max(r[2] for r in rows)
# find the wave start: the first kernel after the longest gap in the last 300 ms... use: last time where a gap > 3 ms precedes, before thi-150ms
prev=None; gaps=[]
for r in rows:
    if r[1] > thi-300e6:
        if prev is not None and r[1]-prev > 2e6: gaps.append((r[1], (r[1]-prev)/1e6))
    prev=max(prev or 0, r[2])
print("gaps>2ms in last 300ms:", [(round((g-thi)/1e6,1), round(d,1)) for g,d in gaps])
t0=[g for g,d in gaps if g < thi-150e6][-1]
print("wave starts at", (t0-thi)/1e6)
n=0
for r in rows:
    if r[1] >= t0 and r[1] < t0+9e6:
        nm=r[0].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][-28:]
        print("%7.3f ms  dur %7.1f us  q%s s%s grid %6d  %s"%((r[1]-t0)/1e6,(r[2]-r[1])/1e3,r[4],r[3],r[5],nm)); n+=1
        if n>140: break
PY
