"""From the traces of probes/trace_hiccup.sh: per batch, the window of its big host-to-device copies, how long after them the
planner's first kernel started, and how long the planner kernels took -- worst cases first."""
import csv
K = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]) for r in csv.DictReader(open('gpurun_out/hic_kernels.csv'))]
C = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'].replace('MEMORY_COPY_', '')) for r in csv.DictReader(open('gpurun_out/hic_copies.csv'))]
K.sort(); C.sort()
reg = [k for k in K if k[2].startswith('k_dplan_region')]
rows = []
for i, r in enumerate(reg):
    prev = reg[i - 1][1] if i else 0
    h2d = [c for c in C if c[2] == 'HOST_TO_DEVICE' and c[1] <= r[0] + 2_000_000 and c[0] >= prev and (c[1] - c[0]) > 200_000]
    if not h2d: continue
    first = min(c[0] for c in h2d); last = max(c[1] for c in h2d)
    comp = [k for k in K if k[2].startswith('k_dplan_compact') and k[0] >= r[0]][0]
    rows.append((i, (last - first) / 1e6, sum(c[1] - c[0] for c in h2d) / 1e6, (r[0] - last) / 1e6, (comp[1] - r[0]) / 1e6))
rows.sort(key=lambda x: -(x[1] + x[3] + x[4]))
for x in rows[:6]:
    print("batch %2d: upload window %.1f ms (engine busy %.1f), planner starts %.1f ms later, takes %.1f ms" % x)
