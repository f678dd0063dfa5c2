import sqlite3, glob, sys
con = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]); cur = con.cursor()
rows=[]
for r in cur.execute("select m.start, m.end, m.size, s.string from rocpd_memory_copy m left join rocpd_string s on m.name_id = s.id where m.size > 1000000"):
    rows.append((r[0], r[1], 'COPY %s %d MiB' % (r[3].replace('MEMORY_COPY_',''), r[2]>>20)))
for r in cur.execute("select start, end, name, grid_x/workgroup_x from kernels where name like '%zmi_%' and (end-start) > 200000"):
    rows.append((r[0], r[1], 'K %s g%d' % (r[2].replace('void ','').split('(')[0][:24], r[3])))
rows.sort()
# the last deflate call: from the last H2D burst backwards
h2d=[i for i,r in enumerate(rows) if 'HOST_TO_DEVICE' in r[2]]
n=int(sys.argv[2])
i0=h2d[-n]
t0=rows[i0][0]
for a,b,nm in rows[i0:]:
    if 'inflate' in nm: break
    print("%8.2f %8.2f %7.2f  %s" % ((a-t0)/1e6, (b-t0)/1e6, (b-a)/1e6, nm))
