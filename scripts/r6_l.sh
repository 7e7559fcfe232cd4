cd /root/repo
python scripts/time_small_fit.py nsf6 10 512 400 2>/dev/null | tail -1
python scripts/time_small_fit.py maf3 10 512 400 2>/dev/null | tail -1
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/rp2 -o t -- python /root/repo/scripts/time_small_fit.py nsf6 10 512 60 > /dev/null 2>&1
python - <<'PY'
import csv, glob
ev=[]
for f in glob.glob('/tmp/rp2/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:50]))
for f in glob.glob('/tmp/rp2/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')+' '+r.get('Bytes', r.get('Size','?'))))
ev.sort()
# the last 2 epochs' worth of events: find the last 3 chain kernels
idx=[i for i,e in enumerate(ev) if 'maf_chain' in e[2]]
i0=idx[-3]; t0=ev[i0][0]
for s,e,n in ev[i0:idx[-1]]:
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  {n}")
PY
