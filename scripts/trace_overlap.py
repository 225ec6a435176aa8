"""Print begin/end timestamps (us, relative) of the kernels of the LAST m17gpu_rx_blocks call from a rocprofv3
--kernel-trace CSV directory:  python scripts/trace_overlap.py <dir>"""
import sys, csv, glob
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
fe = [i for i, r in enumerate(rows) if "k_frontend" in r[2]]
# the last call = the last run of consecutive front-end/timing kernels
last = fe[-1]
start = last
while start > 0 and ("k_frontend" in rows[start - 1][2] or "k_sync_frame" in rows[start - 1][2]) and rows[start][0] - rows[start - 1][0] < 2_000_000:
    start -= 1
t0 = rows[start][0]
for r in rows[start:]:
    print(f"{(r[0]-t0)/1e3:9.1f} .. {(r[1]-t0)/1e3:9.1f} us  q={r[3]:>4s}  {r[2]}")
