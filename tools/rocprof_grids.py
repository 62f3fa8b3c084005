import sqlite3,sys,glob
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt=[t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t][:1] or [t for t in tabs if 'kernel' in t]
print(kt[:5])
v=[t for t in tabs if t=='kernels']
cols=[c[1] for c in db.execute("pragma table_info(kernels)")]
print(cols)
q="select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1000.0 from kernels where name like ? group by name, grid_x, grid_y order by count(*) desc"
for pat in sys.argv[2:]:
    for r in db.execute(q,(pat,)): print(r[0][:70], r[1:])
