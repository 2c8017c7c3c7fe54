import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
st = "start" if "start" in cols else "start_timestamp"; en = "end" if "end" in cols else "end_timestamp"
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f'select name, "{st}", "{en}", grid_x' + (f', {q}' if q else '') + f' from kernels order by "{st}"').fetchall()
rows = rows[len(rows)//2: len(rows)//2 + 24]
t0 = rows[0][1]
for r in rows:
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f} dur {(r[2]-r[1])/1e3:6.1f}  q={r[4] if q else '-'} grid {r[3]:6d}  {r[0][:70]}")
