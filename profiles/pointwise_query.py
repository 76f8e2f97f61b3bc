import sqlite3, sys
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name, grid_x, grid_y, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%pw_%' or name like '%gather_conv%' or name like '%wgrad%' or name like '%pack%' or name like '%reduce_splits%' or name like '%fillBuffer%' group by name, grid_x, grid_y order by name, grid_x").fetchall()
for r in rows: print(f"{r[0][:58]:58s} grid {r[1]:8d} x{r[2]:4d} n={r[3]:3d} avg {r[4]:8.2f} min {r[5]:8.2f}")
