"""Per-kernel MFMA-busy share from a rocprofv3 pass with --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE.
busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): the fraction of SIMD-cycles of
the dispatch in which a matrix instruction was executing (MI355X_MICROARCH.md: the counter counts cycles, GRBM_GUI_ACTIVE
is summed over the 8 XCDs)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
acc = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    a = acc[k]
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        a[0] += float(r['Counter_Value']); a[2] += 1
        a[3] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    elif r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        a[1] += float(r['Counter_Value'])
rows = []
for k, (mf, gui, n, us) in acc.items():
    if mf <= 0 or gui <= 0: continue
    rows.append((us, k, n, mf / (gui / 8 * 256 * 4)))
rows.sort(reverse=True)
print('kernel,dispatches,total_us_in_this_pass,mfma_busy_share')
for us, k, n, share in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f'"{k[:100]}",{n},{us:.0f},{share:.3f}')
