"""Per-kernel HBM traffic from rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch in the CSV).
gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE counts 128-B read requests at 64 B, so it
is doubled; WRITE_SIZE is exact for 16-B/lane stores and float atomics."""
import csv, glob, collections, sys
def load(d, name):
    f = glob.glob(f'{d}/*/*_counter_collection.csv')[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == name:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc
fetch = load(sys.argv[1], 'FETCH_SIZE'); write = load(sys.argv[2], 'WRITE_SIZE')
want = sys.argv[3:] or ['k_ln_apply', 'k_ln_bwd_dense', 'k_adamw', 'k_msda_bwd', 'k_msda_fwd', 'k_window_attn', 'k_add_ln', 'k_sample_select', 'k_mask_loss', 'k_point_sample']
print('kernel,launches,fetch_MB_per_launch(x2 corrected),write_MB_per_launch,total_MB_per_launch')
for k in sorted(set(fetch) | set(write)):
    if not any(w in k for w in want): continue
    f = fetch.get(k, []); w = write.get(k, [])
    fm = 2 * sum(f) / max(1, len(f)) / 1024; wm = sum(w) / max(1, len(w)) / 1024
    print(f'"{k[:90]}",{len(f)},{fm:.1f},{wm:.1f},{fm + wm:.1f}')
